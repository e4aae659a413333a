"""Thin typed wrappers over the UNet-primitive entry points of libmvedit_amd (section 2 of
include/mvedit_amd.h).  Tensors are PyTorch-ROCm tensors used as storage; activations are NHWC
(`[B*H*W, C]` row-major) in fp16 or bf16.  No op here has a torch fallback: if the extension is
missing, importing `mvedit_amd._lib` raises.
"""
import torch

from . import _lib

GEGLU = 1
OUT_F32 = 2
W_CHUNK64 = 4
PAD_BR = 32         # MVE_CONV_PAD_BR: stride-2 conv padded bottom/right only (diffusers Downsample2D(padding=0))

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.int32: 3, torch.uint8: 4}


def dt(t):
    return _DT[t.dtype if isinstance(t, torch.Tensor) else t]


def _s(t):
    return _lib.stream_ptr(t.device)


def _chk16(*ts):
    """16-bit CUDA tensors whose last axis is dense (row-strided views are fine: lda/ldr/ldc are passed)."""
    for t in ts:
        if t is not None:
            assert t.is_cuda and t.stride(-1) == 1 and t.dtype in (torch.float16, torch.bfloat16), (t.dtype, t.shape)


NO_SPLITK = 8


def _chk_lo8(t):
    assert t is None or (t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()), 'a low half is a contiguous uint8 tensor (lo8)'


def split_pair(x32, dtype):
    """fp32 tensor -> (hi, lo8): hi = x rounded to `dtype`, lo8 = E5M2(2^8 (x - hi)) as uint8 -- the stream-pair format of the *_pair entry
    points (include/mvedit_amd.h); torch.float8_e5m2 rounds to nearest even like v_cvt_pk_bf8_f32."""
    hi = x32.to(dtype)
    # clamped to E5M2's finite range and NaN -> 0 like the device packer (common.h: mve_lo8_scaled); gfx950's bf8 is the OCP E5M2 torch.float8_e5m2
    # holds -- on gfx942 the hardware format is FNUZ and these host helpers would not match the device
    r = torch.nan_to_num(((x32.float() - hi.float()) * 256.0), nan=0.0).clamp(-57344.0, 57344.0)
    lo = r.to(torch.float8_e5m2).view(torch.uint8)
    return hi, lo


def lo8_to_float(lo8):
    """uint8 low half of a stream pair -> its fp32 value."""
    return lo8.view(torch.float8_e5m2).float() / 256.0


def _splitk_ws(M, N, K, device, rows_per_image):
    nbytes = _lib.raw('mve_gemm_workspace_bytes')(M, N, K, int(rows_per_image)) if rows_per_image else 0
    return (torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes) if nbytes else (None, 0)


def gemm(a, w, bias=None, rowvec=None, rows_per_vec=0, residual=None, flags=0, out=None, out_scale=1.0, rows_per_image=0, residual_lo=None,
         pair_out=False):
    """a [M,K], w [N,K] -> [M,N] (or [M,N/2] with GEGLU).  bias/rowvec fp32.
    residual_lo / pair_out: the residual-pair entry point (mve_gemm_pair): the residual is residual + lo8_to_float(residual_lo), and with
    pair_out the result comes back as (hi, lo8): hi = round16(v) and the 8-bit low half of v - hi (uint8 tensor, see split_pair / lo8_to_float)."""
    _chk16(a, w, residual)
    _chk_lo8(residual_lo)
    if residual_lo is not None or pair_out:
        M, K = a.shape
        N = w.shape[0]
        out = torch.empty(M, N, dtype=a.dtype, device=a.device)
        out_lo = torch.empty(M, N, dtype=torch.uint8, device=a.device) if pair_out else None
        ws, ws_bytes = _splitk_ws(M, N, K, a.device, rows_per_image)
        with torch.cuda.device(a.device):
            _lib.call('mve_gemm_pair', dt(a), _lib.ptr(a), a.stride(0), _lib.ptr(w), w.stride(0), _lib.ptr(out), out.stride(0),
                      M, N, K, _lib.ptr(bias), _lib.ptr(rowvec), rowvec.stride(0) if rowvec is not None else 0,
                      int(rows_per_vec), _lib.ptr(residual), residual.stride(0) if residual is not None else 0, int(flags), float(out_scale),
                      _lib.ptr(ws), ws_bytes, int(rows_per_image), _lib.ptr(residual_lo), _lib.ptr(out_lo), _s(a))
        return (out, out_lo) if pair_out else out
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and (rowvec is None or rowvec.stride(-1) == 1)
    n_out = N // 2 if flags & GEGLU else N
    if out is None:
        out = torch.empty(M, n_out, dtype=torch.float32 if flags & OUT_F32 else a.dtype, device=a.device)
    ws, ws_bytes = _splitk_ws(M, N, K, a.device, rows_per_image)
    with torch.cuda.device(a.device):
        _lib.call('mve_gemm', dt(a), _lib.ptr(a), a.stride(0), _lib.ptr(w), w.stride(0), _lib.ptr(out), out.stride(0),
                  M, N, K, _lib.ptr(bias), _lib.ptr(rowvec), rowvec.stride(0) if rowvec is not None else 0,
                  int(rows_per_vec), _lib.ptr(residual),
                  residual.stride(0) if residual is not None else 0, int(flags), float(out_scale), _lib.ptr(ws), ws_bytes,
                  int(rows_per_image), _s(a))
    return out


def gemm_ln(a, w, bias, gamma, beta, residual=None, residual_lo=None, pair_out=True, eps=1e-5, rows_per_image=0):
    """mve_gemm_pair_ln: a [M,K], w [N,K] -> ((out, out_lo) or out, LayerNorm(out rows) * gamma + beta).  The LayerNorm reads the row as a consumer
    reads it back (hi + lo8 with pair_out); on the 320-wide pair tile it runs in the producing tile's epilogue, else as the kernel behind the GEMM."""
    _chk16(a, w, residual)
    _chk_lo8(residual_lo)
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    out_lo = torch.empty(M, N, dtype=torch.uint8, device=a.device) if pair_out else None
    ln_out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    ws, ws_bytes = _splitk_ws(M, N, K, a.device, rows_per_image)
    with torch.cuda.device(a.device):
        _lib.call('mve_gemm_pair_ln', dt(a), _lib.ptr(a), a.stride(0), _lib.ptr(w), w.stride(0), _lib.ptr(out), out.stride(0), M, N, K, _lib.ptr(bias),
                  _lib.ptr(residual), residual.stride(0) if residual is not None else 0, _lib.ptr(ws), ws_bytes, int(rows_per_image),
                  _lib.ptr(residual_lo), _lib.ptr(out_lo), _lib.ptr(ln_out), ln_out.stride(0), _lib.ptr(gamma), _lib.ptr(beta), float(eps), _s(a))
    return ((out, out_lo) if pair_out else out), ln_out


def conv3x3(x1, w, B, H, W, x2=None, stride=1, upsample=False, bias=None, rowvec=None, residual=None, flags=0,
            out_scale=1.0, splitk=True, residual_lo=None, pair_out=False):
    """x1 [B*H*W, C1] (+ x2 [B*H*W, C2]) NHWC, w [Cout,3,3,C1+C2] -> ([B*Ho*Wo, Cout], Ho, Wo).
    With flags & W_CHUNK64 the weight is [Cout, (C1+C2)/64, 3, 3, 64] (see pack_conv_weight).
    residual_lo / pair_out: the residual-pair entry point (mve_conv3x3_pair) -> ((out, out_lo), Ho, Wo) with pair_out."""
    _chk16(x1, x2, w, residual)
    _chk_lo8(residual_lo)
    C1 = x1.shape[1]
    C2 = x2.shape[1] if x2 is not None else 0
    Cout = w.shape[0]
    assert w.numel() == Cout * 9 * (C1 + C2)
    Hv, Wv = (H * 2, W * 2) if upsample else (H, W)
    Ho, Wo = (Hv - 1) // stride + 1, (Wv - 1) // stride + 1
    out = torch.empty(B * Ho * Wo, Cout, dtype=torch.float32 if flags & OUT_F32 else x1.dtype, device=x1.device)
    ws, ws_bytes = _splitk_ws(B * Ho * Wo, Cout, 9 * (C1 + C2), x1.device, Ho * Wo if splitk else 0)
    if residual_lo is not None or pair_out:
        out_lo = torch.empty(out.shape, dtype=torch.uint8, device=out.device) if pair_out else None
        with torch.cuda.device(x1.device):
            _lib.call('mve_conv3x3_pair', dt(x1), _lib.ptr(x1), C1, _lib.ptr(x2), C2, B, H, W, stride, int(bool(upsample)),
                      _lib.ptr(w), Cout, _lib.ptr(out), out.stride(0), _lib.ptr(bias), _lib.ptr(rowvec),
                      rowvec.stride(0) if rowvec is not None else 0, _lib.ptr(residual),
                      residual.stride(0) if residual is not None else 0, int(flags), float(out_scale), _lib.ptr(ws), ws_bytes,
                      _lib.ptr(residual_lo), _lib.ptr(out_lo), _s(x1))
        return ((out, out_lo) if pair_out else out), Ho, Wo
    with torch.cuda.device(x1.device):
        _lib.call('mve_conv3x3', dt(x1), _lib.ptr(x1), C1, _lib.ptr(x2), C2, B, H, W, stride, int(bool(upsample)),
                  _lib.ptr(w), Cout, _lib.ptr(out), out.stride(0), _lib.ptr(bias), _lib.ptr(rowvec),
                  rowvec.stride(0) if rowvec is not None else 0, _lib.ptr(residual),
                  residual.stride(0) if residual is not None else 0, int(flags), float(out_scale), _lib.ptr(ws), ws_bytes, _s(x1))
    return out, Ho, Wo


def conv3x3_shortcut(x1, w, B, H, W, x3, x4=None, bias=None, bias2=None, residual=None, splitk=True, pair_out=False):
    """ResnetBlock2D tail in one launch: conv3x3(x1) + conv1x1(cat[x3, x4]) + bias + bias2 (+ residual).
    w [Cout, 9*C1 + C3 + C4]: 3x3 part in the channel-slab-major order of pack_conv_weight(..., True), then the 1x1 matrix.
    pair_out (no residual): mve_conv3x3_shortcut_pair -> (out, out_lo)."""
    _chk16(x1, x3, x4, w, residual)
    C1, C3 = x1.shape[1], x3.shape[1]
    C4 = x4.shape[1] if x4 is not None else 0
    Cout = w.shape[0]
    assert w.numel() == Cout * (9 * C1 + C3 + C4)
    out = torch.empty(B * H * W, Cout, dtype=x1.dtype, device=x1.device)
    ws, ws_bytes = _splitk_ws(B * H * W, Cout, 9 * C1 + C3 + C4, x1.device, H * W if splitk else 0)
    if pair_out:
        assert residual is None
        out_lo = torch.empty(out.shape, dtype=torch.uint8, device=out.device)
        with torch.cuda.device(x1.device):
            _lib.call('mve_conv3x3_shortcut_pair', dt(x1), _lib.ptr(x1), C1, _lib.ptr(x3), C3, _lib.ptr(x4), C4, B, H, W, _lib.ptr(w), Cout,
                      _lib.ptr(out), out.stride(0), _lib.ptr(bias), _lib.ptr(bias2), 0, 1.0, _lib.ptr(ws), ws_bytes, _lib.ptr(out_lo), _s(x1))
        return out, out_lo
    with torch.cuda.device(x1.device):
        _lib.call('mve_conv3x3_shortcut', dt(x1), _lib.ptr(x1), C1, _lib.ptr(x3), C3, _lib.ptr(x4), C4, B, H, W, _lib.ptr(w), Cout,
                  _lib.ptr(out), out.stride(0), _lib.ptr(bias), _lib.ptr(bias2), _lib.ptr(residual),
                  residual.stride(0) if residual is not None else 0, 0, 1.0, _lib.ptr(ws), ws_bytes, _s(x1))
    return out


def pack_upsample_phase_weights(w_oihw, dtype=None):
    """torch conv weight [O, I, 3, 3] of an Upsample2D conv (I % 64 == 0, on the GPU) -> the summed taps of its four 2 x 2 phase convs,
    [4, O, I/64, 2, 2, 64] in `dtype` (default: the weight's), see mve_upsample_conv_phases."""
    O, I = w_oihw.shape[:2]
    dtype = dtype or w_oihw.dtype
    w = w_oihw.contiguous()
    w4 = torch.empty(4, O, I // 64, 2, 2, 64, dtype=dtype, device=w.device)
    with torch.cuda.device(w.device):
        _lib.call('mve_pack_upsample_phase_weights', _DT[w.dtype], _DT[dtype], _lib.ptr(w), O, I, _lib.ptr(w4), _s(w))
    return w4


def upsample_conv_phases_supported(C, Cout, B, H, W):
    return bool(_lib.raw('mve_upsample_conv_phases_supported')(C, Cout, B, H, W))


def upsample_conv_phases(x, w4, B, H, W, bias=None, pair_out=False, splitk=True):
    """x [B*H*W, C] NHWC, w4 from pack_upsample_phase_weights -> nearest-2x upsample + conv3x3 (pad 1) as [B*2H*2W, Cout] (, low half)."""
    _chk16(x, w4)
    C, Cout = x.shape[1], w4.shape[1]
    assert w4.numel() == 16 * Cout * C
    out = torch.empty(B * 4 * H * W, Cout, dtype=x.dtype, device=x.device)
    out_lo = torch.empty(out.shape, dtype=torch.uint8, device=out.device) if pair_out else None
    nb = _lib.raw('mve_upsample_conv_phases_workspace_bytes')(C, Cout, B, H, W) if splitk else 0
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device) if nb else None
    with torch.cuda.device(x.device):
        _lib.call('mve_upsample_conv_phases', dt(x), _lib.ptr(x), C, B, H, W, _lib.ptr(w4), Cout, _lib.ptr(out), _lib.ptr(bias), 0,
                  _lib.ptr(ws), nb, _lib.ptr(out_lo), _s(x))
    return (out, out_lo) if pair_out else out


def pack_conv_weight(w_oihw, chunk64=None):
    """torch conv weight [O, I, 3, 3] -> (packed weight, flag): [O,3,3,I], or [O, I/64, 3, 3, 64] when I % 64 == 0."""
    O, I = w_oihw.shape[:2]
    if chunk64 is None:
        chunk64 = I % 64 == 0
    if chunk64:
        return w_oihw.reshape(O, I // 64, 64, 3, 3).permute(0, 1, 3, 4, 2).contiguous(), W_CHUNK64
    return w_oihw.permute(0, 2, 3, 1).contiguous(), 0


def groupnorm(x1, B, HW, gamma, beta, G=32, eps=1e-5, silu=True, x2=None):
    _chk16(x1, x2)
    C1 = x1.shape[1]
    C2 = x2.shape[1] if x2 is not None else 0
    C = C1 + C2
    out = torch.empty(B * HW, C, dtype=x1.dtype, device=x1.device)
    ws = torch.empty(_lib.raw('mve_groupnorm_workspace_bytes')(B, HW, C, G), dtype=torch.uint8, device=x1.device)
    with torch.cuda.device(x1.device):
        _lib.call('mve_groupnorm_silu', dt(x1), _lib.ptr(x1), C1, _lib.ptr(x2), C2, B, HW, G, float(eps), _lib.ptr(gamma),
                  _lib.ptr(beta), int(bool(silu)), _lib.ptr(out), _lib.ptr(ws), _s(x1))
    return out


def layernorm(x, gamma, beta, eps=1e-5):
    _chk16(x)
    M, C = x.shape
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.call('mve_layernorm', dt(x), _lib.ptr(x), x.stride(0), _lib.ptr(y), y.stride(0), M, C, _lib.ptr(gamma),
                  _lib.ptr(beta), float(eps), _s(x))
    return y


def attention(q, k, v, B, Lq, Lk, heads, head_dim, scale=None, k2=None, v2=None, Lk2=0, out=None, prescaled=False):
    """q: [B*Lq, >=heads*d] view (any row stride), k/v: [B*Lk, ...] views; returns [B*Lq, heads*d].
    prescaled: q already carries softmax_scale * log2(e) (mve_attention_prescaled)."""
    for t in (q, k, v, k2, v2):
        if t is not None:
            assert t.is_cuda and t.stride(1) == 1 and t.dtype in (torch.float16, torch.bfloat16)
    if scale is None:
        scale = head_dim ** -0.5
    if out is None:
        out = torch.empty(B * Lq, heads * head_dim, dtype=q.dtype, device=q.device)
    with torch.cuda.device(q.device):
        args = (dt(q), _lib.ptr(q), q.stride(0), _lib.ptr(k), k.stride(0), _lib.ptr(v), v.stride(0),
                _lib.ptr(k2), k2.stride(0) if k2 is not None else 0, _lib.ptr(v2), v2.stride(0) if v2 is not None else 0,
                _lib.ptr(out), out.stride(0), B, Lq, Lk, int(Lk2), heads, head_dim)
        if prescaled:
            _lib.call('mve_attention_prescaled', *args, _s(q))
        else:
            _lib.call('mve_attention', *args, float(scale), _s(q))
    return out


def nchw_to_nhwc(x, dtype, cpad=None):
    B, C, H, W = x.shape
    cpad = cpad or (C + 7) // 8 * 8
    x = x.contiguous()
    y = torch.empty(B * H * W, cpad, dtype=dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call('mve_nchw_to_nhwc', dt(dtype), dt(x), _lib.ptr(x), B, C, H, W, cpad, _lib.ptr(y), _s(x))
    return y


def nhwc_to_nchw(x, B, C, H, W, dtype):
    assert x.is_contiguous() or x.stride(1) == 1
    y = torch.empty(B, C, H, W, dtype=dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call('mve_nhwc_to_nchw', dt(dtype), dt(x), _lib.ptr(x), x.stride(0), B, C, H, W, _lib.ptr(y), _s(x))
    return y


def timestep_embedding(t, dim, dtype):
    t = t.float().contiguous()
    out = torch.empty(t.shape[0], dim, dtype=dtype, device=t.device)
    with torch.cuda.device(t.device):
        _lib.call('mve_timestep_embedding', dt(dtype), _lib.ptr(t), t.shape[0], dim, _lib.ptr(out), _s(t))
    return out


def silu(x):
    _chk16(x)
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.call('mve_silu', dt(x), _lib.ptr(x), _lib.ptr(y), x.numel(), _s(x))
    return y


def axpy(a, b, alpha=1.0):
    _chk16(a, b)
    y = torch.empty_like(a)
    with torch.cuda.device(a.device):
        _lib.call('mve_axpy', dt(a), _lib.ptr(a), _lib.ptr(b), float(alpha), _lib.ptr(y), a.numel(), _s(a))
    return y


def cfg_combine(uncond, text, guidance_scale):
    uncond = uncond.float().contiguous()
    text = text.float().contiguous()
    out = torch.empty_like(uncond)
    with torch.cuda.device(uncond.device):
        _lib.call('mve_cfg_combine', _lib.ptr(uncond), _lib.ptr(text), float(guidance_scale), _lib.ptr(out),
                  uncond.numel(), _s(uncond))
    return out


def softmax_rows(scores, dtype):
    """Row softmax of fp32 scores [M, N] into 16-bit probabilities (the VAE mid-block attention between its two GEMMs)."""
    assert scores.dtype == torch.float32 and scores.dim() == 2 and scores.is_contiguous()
    M, N = scores.shape
    out = torch.empty(M, N, dtype=dtype, device=scores.device)
    with torch.cuda.device(scores.device):
        _lib.call('mve_softmax_rows', dt(dtype), _lib.ptr(scores), N, M, N, _lib.ptr(out), N, _s(scores))
    return out
