"""LPIPS-VGG patch loss, forward and backward (csrc/lpips.hip + the executor's LPIPS mode behind mvedit_amd.lpips.LPIPSEngine) vs the
torch restatement of lpips==0.1.4 with torch autograd for the gradient (oracle/lpips_oracle.py; unpinned: the package is absent).
SURVEY section 8(f) rank 1 (image-space loss of the reconstruct step)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import lpips_oracle as L


# ------------------------------------------------------------------------------------------------ CPU
def test_plan_and_inventory(lib):
    """Plan-time only: op counts, the forward / backward split, conv FLOPs (VGG16 at 128^2: 5.02 GMAC per image forward; the backward
    adds every conv's dgrad on the prediction half), the parameter inventory equal to lpips' state dict."""
    import ctypes
    from mvedit_amd import _lib
    from mvedit_amd.lpips import LPIPSEngine
    eng = LPIPSEngine(torch.bfloat16, 'cpu')
    info = eng.plan(8, 128, 128)
    macs = sum(ci * co * 9 * (128 >> s) ** 2 for (ci, co), s in zip(L.VGG_CH, (0, 0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4)))
    assert abs(macs / 1e9 - 5.02) < 0.02
    pad = (8 - 3) * 64 * 9 * 128 * 128                                  # conv1_1 runs on 8 padded input channels, its dgrad on 8 padded outputs
    want = 2 * (16 * (macs + pad) + 8 * (macs + pad))
    assert abs(info['conv_flops'] - want) < 1e-6 * want, (info['conv_flops'], want)
    assert info['n_forward_ops'] == 1 + 13 * 2 + 4 + 5 and info['n_ops'] == info['n_forward_ops'] + 5 + 4 + 13 * 2 + 4 + 1
    buf = ctypes.create_string_buffer(256)
    assert _lib.raw('mve_unet_missing_params')(eng._h, buf, 256) == len(L.param_shapes())
    with pytest.raises(_lib.MveError):
        eng.plan(1, 100, 128)                                               # not divisible by 16


def test_oracle_properties():
    """The restatement behaves like a distance: zero for identical images, positive otherwise, invariant to the batch composition;
    its lin / normalisation wiring equals a direct evaluation of the published formula on one layer."""
    sd = L.random_params(1)
    g = torch.Generator().manual_seed(0)
    a, b = torch.rand(3, 3, 32, 32, generator=g), torch.rand(3, 3, 32, 32, generator=g)
    with torch.no_grad():
        d = L.lpips(sd, a, b)
        assert d.shape == (3,) and (d > 0).all() and L.lpips(sd, a, a).abs().max() == 0
        assert torch.allclose(L.lpips(sd, a[1:2], b[1:2]), d[1:2], rtol=1e-5)
        f0, f1 = L.features(sd, a * 2 - 1), L.features(sd, b * 2 - 1)
        u0, u1 = F.normalize(f0[0], dim=1, eps=0) , F.normalize(f1[0], dim=1, eps=0)
        first = ((u0 - u1) ** 2 * sd['lin0.model.1.weight']).sum(1).mean(dim=(1, 2))
        rest = sum(F.conv2d((F.normalize(x, dim=1, eps=0) - F.normalize(y, dim=1, eps=0)) ** 2, sd[f'lin{k}.model.1.weight']).mean(dim=(2, 3)).flatten()
                   for k, (x, y) in enumerate(zip(f0, f1)) if k > 0)
        assert torch.allclose(first + rest, d, rtol=1e-4)


# ------------------------------------------------------------------------------------------------ GPU
def _case(B=3, S=32, seed=2):
    sd = L.random_params(seed)
    g = torch.Generator().manual_seed(seed + 1)
    return sd, torch.rand(B, 3, S, S, generator=g), torch.rand(B, 3, S, S, generator=g)


@pytest.mark.gpu
def test_forward_and_gradient_vs_oracle(lib):
    """bf16, the reference's configuration (lpips_loss.py:31).  Bar: PyTorch's own bf16 module + autograd sits 5e-3 (loss) / 6.6e-2
    rel-L2 (gradient, cosine 0.998) away from fp32 on this case (measured on the CPU, recomputed below); the engine must be at least
    that close to fp32 (x1.25 + a small absolute slack), and close to the bf16 module itself."""
    from mvedit_amd.lpips import LPIPSEngine
    dtype = torch.bfloat16
    sd, pred, target = _case()
    sdq = {k: v.to(dtype).float() for k, v in sd.items()}
    coef = torch.tensor([1.0, 0.5, 2.0])
    p32 = pred.to(dtype).float().requires_grad_(True)
    d32 = L.lpips(sdq, p32, target.to(dtype).float())
    (d32 * coef).sum().backward()
    ph = pred.to(dtype).requires_grad_(True)
    dh = L.lpips_half(sdq, ph, target, dtype)
    (dh * coef).sum().backward()
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    emu_loss, emu_grad = ((dh.detach() - d32.detach()).abs() / d32.detach()).max().item(), rel(ph.grad, p32.grad)
    eng = LPIPSEngine.from_state_dict(sdq, dtype)
    pg = pred.to(dtype).float().cuda().requires_grad_(True)
    d = eng(pg, target.to(dtype).float().cuda())
    assert d.shape == (3,) and d.dtype == torch.float32
    (d * coef.cuda()).sum().backward()
    err_loss = ((d.cpu() - d32.detach()).abs() / d32.detach()).max().item()
    err_grad = rel(pg.grad.cpu(), p32.grad)
    cos = F.cosine_similarity(pg.grad.cpu().flatten(), p32.grad.flatten(), dim=0).item()
    print(f'loss: engine {err_loss:.2e} vs torch-bf16 {emu_loss:.2e};  gradient rel-L2: engine {err_grad:.2e} vs torch-bf16 {emu_grad:.2e}, cosine {cos:.5f}')
    assert err_loss <= 1.25 * emu_loss + 2e-3
    assert err_grad <= 1.25 * emu_grad + 1e-2 and cos > 0.995
    assert torch.equal(eng(pg.detach(), pg.detach()).cpu(), torch.zeros(3))          # identical images: exactly zero


@pytest.mark.gpu
def test_fp16_forward_and_patch_size(lib):
    """fp16 engine, forward only (fp16 autograd underflows in the reference's stack as well: 19 % off fp32), at the production patch
    size 8 x 128 x 128; batch invariance of the per-pair values."""
    from mvedit_amd.lpips import LPIPSEngine
    sd, pred, target = _case(B=8, S=128, seed=4)
    sdq = {k: v.half().float() for k, v in sd.items()}
    with torch.no_grad():
        d32 = L.lpips(sdq, pred.half().float(), target.half().float())
    eng = LPIPSEngine.from_state_dict(sdq, torch.float16)
    with torch.no_grad():
        d = eng(pred.half().cuda(), target.half().cuda())
        assert ((d.cpu() - d32).abs() / d32).max() < 5e-3
        assert torch.equal(eng(pred[2:5].half().cuda(), target[2:5].half().cuda()), d[2:5])


@pytest.mark.gpu
def test_lpips_building_blocks(lib):
    """max pooling (+ backward with torch's first-arg-max rule), ReLU backward and one layer's distance + gradient, each against torch."""
    import ctypes
    from mvedit_amd import _lib
    from mvedit_amd.ops import dt as _dt
    dtype = torch.float16
    g = torch.Generator().manual_seed(5)
    B, H, W, C = 2, 8, 12, 64
    x = torch.randn(B, C, H, W, generator=g).to(dtype)
    x[:, :, :2, :2] = 0.5                                             # ties inside a window
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    xr = x.float().requires_grad_(True)
    y_ref = F.max_pool2d(xr, 2, 2)
    gy = torch.randn(y_ref.shape, generator=g).to(dtype)
    y_ref.backward(gy.float())
    xd, gyd = nhwc(x), nhwc(gy)
    y, gx = torch.empty(B, H // 2, W // 2, C, dtype=dtype, device='cuda'), torch.empty(B, H, W, C, dtype=dtype, device='cuda')
    s = _lib.stream_ptr(xd.device)
    _lib.call('mve_maxpool2x2', _dt(dtype), _lib.ptr(xd), B, H, W, C, _lib.ptr(y), s)
    _lib.call('mve_maxpool2x2_backward', _dt(dtype), _lib.ptr(xd), _lib.ptr(gyd), B, H, W, C, _lib.ptr(gx), s)
    assert torch.equal(y.cpu().permute(0, 3, 1, 2).float(), y_ref.detach())
    assert torch.equal(gx.cpu().permute(0, 3, 1, 2).float(), xr.grad)
    # relu backward
    a = torch.relu(x)
    ad, gd = nhwc(a), nhwc(torch.ones_like(a))
    _lib.call('mve_relu_backward', _dt(dtype), _lib.ptr(gd), _lib.ptr(ad), gd.numel(), s)
    assert torch.equal(gd.cpu().permute(0, 3, 1, 2).float(), (a > 0).float())
    # one layer: feat = [pred half | target half]
    f = torch.relu(torch.randn(2 * B, C, H, W, generator=g)).to(dtype)
    w = torch.rand(C, generator=g)
    fr = f.float().requires_grad_(True)
    u = fr / (torch.sqrt((fr ** 2).sum(1, keepdim=True)) + 1e-10)
    val = (((u[:B] - u[B:]) ** 2) * w.view(1, C, 1, 1)).sum(1).mean(dim=(1, 2))
    coef = torch.tensor([0.7, 1.3])
    (val * coef).sum().backward()
    fd, wd, cd = nhwc(f), w.cuda(), coef.cuda()
    loss = torch.zeros(B, device='cuda')
    scratch = torch.empty(_lib.raw('mve_lpips_layer_scratch_bytes')(B, H * W), dtype=torch.uint8, device='cuda')
    _lib.call('mve_lpips_layer', _dt(dtype), _lib.ptr(fd), _lib.ptr(wd), B, H * W, C, 0, _lib.ptr(loss), _lib.ptr(scratch), s)
    gf = torch.empty(B, H, W, C, dtype=dtype, device='cuda')
    _lib.call('mve_lpips_layer_backward', _dt(dtype), _lib.ptr(fd), _lib.ptr(wd), _lib.ptr(cd), B, H * W, C, _lib.ptr(gf), s)
    assert torch.allclose(loss.cpu(), val.detach(), rtol=1e-5, atol=1e-7)
    ref = fr.grad[:B]
    assert ((gf.cpu().permute(0, 3, 1, 2).float() - ref).norm() / ref.norm()) < 2e-3


def test_dgrad_weight_packing_formula():
    """The loader packs every VGG weight a second time so that the FORWARD implicit-GEMM kernel computes the data gradient:
    W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx] in the slab-major layout [O'][I'/64][9][64].  This replays the loader's PackDims
    (csrc/unet.hip, LPIPS branch of load_param: source offset 8 + o'*9 + slab*64*ci*9 - tap + c*ci*9) in numpy, reads the result back
    with the forward layout's meaning, and checks conv2d(g, W') against autograd's input gradient."""
    import numpy as np
    rng = np.random.default_rng(0)
    co, ci = 128, 64
    W = rng.standard_normal((co, ci, 3, 3)).astype(np.float32)
    D, s, t = (ci, co // 64, 9, 64), (9, 64 * ci * 9, -1, ci * 9), (9 * co, 9 * 64, 64, 1)
    idx = np.indices(D)
    dst = np.zeros(ci * 9 * co, np.float32)
    dst[sum(idx[k] * t[k] for k in range(4)).reshape(-1)] = W.reshape(-1)[(8 + sum(idx[k] * s[k] for k in range(4))).reshape(-1)]
    Wp = dst.reshape(ci, co // 64, 3, 3, 64).transpose(0, 1, 4, 2, 3).reshape(ci, co, 3, 3).copy()
    x = torch.randn(2, ci, 6, 5, requires_grad=True)
    y = F.conv2d(x, torch.from_numpy(W), padding=1)
    g = torch.randn(y.shape, generator=torch.Generator().manual_seed(1))
    y.backward(g)
    assert torch.allclose(F.conv2d(g, torch.from_numpy(Wp), padding=1), x.grad, rtol=1e-4, atol=1e-4)


def test_layer_gradient_closed_form():
    """The per-pixel closed form the layer-backward kernel evaluates (csrc/lpips.hip, k_lpips_layer<MODE 1>):
        g_c = coef (2 w_c d_c / n0 - (sum_k 2 w_k d_k f0_k) f0_c / (n0^2 |f0|)),  d = f0 / n0 - f1 / n1,  n = |f| + 1e-10,  coef = dL/dloss / HW
    against autograd over lpips' normalize_tensor / lin / spatial-average wiring, float64."""
    g = torch.Generator().manual_seed(7)
    B, C, H, W = 2, 64, 5, 6
    f0 = torch.relu(torch.randn(B, C, H, W, generator=g, dtype=torch.float64)).requires_grad_(True)
    f1 = torch.relu(torch.randn(B, C, H, W, generator=g, dtype=torch.float64))
    w = torch.rand(C, generator=g, dtype=torch.float64)
    coef = torch.tensor([0.7, 1.3], dtype=torch.float64)
    u0 = f0 / (torch.sqrt((f0 ** 2).sum(1, keepdim=True)) + 1e-10)
    u1 = f1 / (torch.sqrt((f1 ** 2).sum(1, keepdim=True)) + 1e-10)
    val = (((u0 - u1) ** 2) * w.view(1, C, 1, 1)).sum(1).mean(dim=(1, 2))
    (val * coef).sum().backward()
    x, y = f0.detach(), f1
    r0 = torch.sqrt((x ** 2).sum(1, keepdim=True))
    i0, i1 = 1 / (r0 + 1e-10), 1 / (torch.sqrt((y ** 2).sum(1, keepdim=True)) + 1e-10)
    d = x * i0 - y * i1
    wv = w.view(1, C, 1, 1)
    dot = (2 * wv * d * x).sum(1, keepdim=True)
    k2 = dot * i0 * i0 / r0
    closed = (coef.view(B, 1, 1, 1) / (H * W)) * (2 * wv * d * i0 - k2 * x)
    assert (closed - f0.grad).abs().max() < 1e-12


def test_backward_schedule_equals_autograd():
    """The executor's backward op order (csrc/unet.hip, build_lpips: per block from the deepest -- layer gradient (closed form), add to
    the running gradient, then per conv in reverse: ReLU mask on the saved output, dgrad = conv with the transposed / flipped weight,
    and a first-arg-max pool backward between blocks) replayed with torch ops in fp32 on the CPU, against autograd over the oracle."""
    sd = L.random_params(3)
    g = torch.Generator().manual_seed(9)
    pred, target = torch.rand(2, 3, 32, 32, generator=g), torch.rand(2, 3, 32, 32, generator=g)
    coef = torch.tensor([1.0, 0.5])
    p = pred.clone().requires_grad_(True)
    (L.lpips(sd, p, target) * coef).sum().backward()
    B = 2
    x = torch.cat([pred, target]) * 2 - 1
    h = (x - sd['scaling_layer.shift']) / sd['scaling_layer.scale']
    acts, pools = [], {}
    for i in range(13):
        if i in (2, 4, 7, 10):
            h = F.max_pool2d(h, 2, 2)
        n = f'net.slice{L.VGG_SLICE[i]}.{L.VGG_IDX[i]}'
        h = F.relu(F.conv2d(h, sd[f'{n}.weight'], sd[f'{n}.bias'], padding=1))
        acts.append(h)
    first = (0, 2, 4, 7, 10, 13)

    def layer_grad(f, w, k):
        x0, y = f[:B], f[B:]
        r0 = torch.sqrt((x0 ** 2).sum(1, keepdim=True))
        i0, i1 = 1 / (r0 + 1e-10), 1 / (torch.sqrt((y ** 2).sum(1, keepdim=True)) + 1e-10)
        d = x0 * i0 - y * i1
        wv = w.view(1, -1, 1, 1)
        k2 = (2 * wv * d * x0).sum(1, keepdim=True) * i0 * i0 / r0
        return (coef.view(B, 1, 1, 1) / (f.shape[2] * f.shape[3])) * (2 * wv * d * i0 - k2 * x0)

    def pool_bwd(xin, gy):                                   # first arg-max in row-major window order
        Bn, C, H, W = xin.shape
        win = xin.reshape(Bn, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(Bn, C, H // 2, W // 2, 4)
        arg = win.argmax(-1)                                 # torch.argmax returns the first maximal index
        gx = torch.zeros_like(win).scatter_(-1, arg[..., None], gy[..., None])
        return gx.reshape(Bn, C, H // 2, W // 2, 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(Bn, C, H, W)

    gcur = None
    for k in range(4, -1, -1):
        last = first[k + 1] - 1
        gf = layer_grad(acts[last], sd[f'lin{k}.model.1.weight'].flatten(), k)
        gcur = gf if gcur is None else gcur + gf
        for i in range(last, first[k] - 1, -1):
            gcur = gcur * (acts[i][:B] > 0)
            wt = sd[f'net.slice{L.VGG_SLICE[i]}.{L.VGG_IDX[i]}.weight']
            gcur = F.conv2d(gcur, wt.flip(2, 3).transpose(0, 1), padding=1)
        if k > 0:
            gcur = pool_bwd(acts[first[k] - 1][:B], gcur)
    grad = gcur / sd['scaling_layer.scale'] * 2
    assert ((grad - p.grad).norm() / p.grad.norm()) < 1e-5
