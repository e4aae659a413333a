"""Host-side mirror of the reference's UNet seam on top of the native executor (csrc/unet.hip).

`UNet2DConditionEngine` is called exactly like the diffusers `UNet2DConditionModel` the reference
pipelines hold as `self.unet` (lib/pipelines/adapter3d_mixin.py:117-125):

    unet(sample, t, encoder_hidden_states=..., cross_attention_kwargs=dict(num_cross_attn_imgs=2),
         down_block_additional_residuals=[...12...], mid_block_additional_residual=..., return_dict=False)[0]

and `unet_enc(unet, ...)` / `unet_dec(unet, ...)` below mirror lib/models/architecture/diffusers.py:57-164.
PyTorch tensors are storage only: one C-ABI call runs the whole forward on the current HIP stream.
There is no torch fallback -- construction fails if libmvedit_amd.so is missing.
"""
import ctypes
import os
from types import SimpleNamespace

import torch

from . import _lib
from .ops import dt as _dt

SD15_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    down_attn=(True, True, True, False), num_heads=(8, 8, 8, 8), cross_attention_dim=768, norm_num_groups=32,
    norm_eps=1e-5, transformer_layers=(1, 1, 1, 1), use_linear_projection=False)

# Stable Diffusion 2.1 topology (the Zero123++ base model, lib/pipelines/zero123plus.py): head_dim 64, linear projections, ctx 1024
SD21_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    down_attn=(True, True, True, False), num_heads=(5, 10, 20, 20), cross_attention_dim=1024, norm_num_groups=32,
    norm_eps=1e-5, transformer_layers=(1, 1, 1, 1), use_linear_projection=True)

OP_CLASSES = ('conv3x3', 'linear', 'attention', 'norm', 'other')


def config_from_diffusers(cfg):
    """Translate a diffusers UNet2DConditionModel config (dict / FrozenDict) into the engine's topology dict."""
    g = cfg.get if hasattr(cfg, 'get') else (lambda k, d=None: getattr(cfg, k, d))
    ch = tuple(g('block_out_channels'))
    n = len(ch)
    heads = g('num_attention_heads') or g('attention_head_dim')      # diffusers 0.27.2 unet_2d_condition.py naming quirk
    heads = tuple(heads) if isinstance(heads, (list, tuple)) else (heads,) * n
    tl = g('transformer_layers_per_block', 1)
    tl = tuple(tl) if isinstance(tl, (list, tuple)) else (tl,) * n
    return dict(in_channels=g('in_channels'), out_channels=g('out_channels'), block_out_channels=ch,
                layers_per_block=g('layers_per_block'), down_attn=tuple('CrossAttn' in t for t in g('down_block_types')),
                num_heads=heads, cross_attention_dim=g('cross_attention_dim'), norm_num_groups=g('norm_num_groups'),
                norm_eps=g('norm_eps'), transformer_layers=tl, use_linear_projection=bool(g('use_linear_projection', False)))


class UNet2DConditionEngine:
    def __init__(self, config=None, dtype=torch.float16, device='cuda'):
        self.cfg = dict(config or SD15_CONFIG)
        assert dtype in (torch.float16, torch.bfloat16)
        self.dtype = dtype
        self.device = torch.device(device)
        c = self.cfg
        n = len(c['block_out_channels'])
        arr = lambda xs: (ctypes.c_int * n)(*[int(x) for x in xs])
        self._h = ctypes.c_void_p()
        _lib.call('mve_unet_create', ctypes.byref(self._h), _dt(dtype), c['in_channels'], c['out_channels'], n,
                  arr(c['block_out_channels']), c['layers_per_block'], arr(c['down_attn']), arr(c['num_heads']),
                  arr(c['transformer_layers']), c['cross_attention_dim'], c['norm_num_groups'], float(c['norm_eps']),
                  int(c['use_linear_projection']))
        self._ws = None
        self._ip = (0, 1.0)          # IP-Adapter: (num_tokens, scale); see set_ip_adapter
        self._ref_keep = None
        # the attributes the reference's pipelines / runner read from a diffusers model
        self.config = SimpleNamespace(in_channels=c['in_channels'], out_channels=c['out_channels'], sample_size=64,
                                      center_input_sample=False, addition_embed_type=None, **{
                                          'block_out_channels': c['block_out_channels'],
                                          'cross_attention_dim': c['cross_attention_dim']})

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            try:
                _lib.raw('mve_unet_destroy')(h)
            except Exception:
                pass
            self._h = None

    def set_residual_pair(self, flag=True):
        """Carry the residual stream (x + f(x) of ResnetBlock2D / BasicTransformerBlock / Transformer2DModel) as an unrounded pair -- the 16-bit tensor
        every matrix-core operand read sees plus an 8-bit E5M2 remainder (lo8, csrc/common.h) -- instead of rounding it after every block as the
        reference's half modules do (mve_unet_set_residual_mode): the end-to-end error against fp32 arithmetic falls below north_star's 1e-3
        (1.23e-3 without) for 1 more byte per stream element written and read.  ON by default for the UNet since round 5 (the native handle is
        created in this mode; MVE_RESIDUAL_PAIR=0 in the environment -- read by UNet handles only -- or set_residual_pair(False) give the reference's
        rounding points); a ControlNetEngine keeps the 16-bit stream unless its own set_residual_pair(True) is called.  Returns the previous setting."""
        return bool(_lib.raw('mve_unet_set_residual_mode')(self._h, int(bool(flag))))

    @property
    def residual_pair(self):
        return bool(_lib.raw('mve_unet_set_residual_mode')(self._h, -1))

    def enable_graph(self, flag=True):
        """Opt-in hipGraph replay of forwards whose plan and tensors (addresses) repeat -- for launch-bound small batches (mve_unet_graph)."""
        return bool(_lib.raw('mve_unet_graph')(self._h, int(bool(flag))))

    # ------------------------------------------------------------------ weights
    @classmethod
    def from_state_dict(cls, state_dict, config=None, dtype=torch.float16, device='cuda'):
        eng = cls(config, dtype, device)
        eng.load_state_dict(state_dict)
        return eng

    def load_state_dict(self, state_dict, strict=True):
        """Copy a diffusers-format state dict into engine-owned packed device storage."""
        with torch.cuda.device(self.device):
            s = _lib.stream_ptr(self.device)
            for name, t in state_dict.items():
                t = t.detach()
                if t.dtype not in (torch.float32, torch.float16, torch.bfloat16):
                    t = t.float()
                t = t.to(self.device).contiguous()
                shape = (ctypes.c_longlong * t.dim())(*t.shape)
                _lib.call('mve_unet_load_param', self._h, name.encode(), _lib.ptr(t), _dt(t), t.dim(), shape, s)
            torch.cuda.current_stream(self.device).synchronize()   # source tensors may die after return
        if strict:
            buf = ctypes.create_string_buffer(256)
            missing = _lib.raw('mve_unet_missing_params')(self._h, buf, 256)
            if missing:
                raise KeyError(f'{missing} UNet parameters missing from the state dict (first: {buf.value.decode()})')
        return self

    # ------------------------------------------------------------------ attention processors
    _REF_KEY = '__mvedit_amd_ref_store__'

    def set_ip_adapter(self, num_tokens=16, scale=1.0):
        """Install / remove (num_tokens=0) the IPAttnProcessor2_0 behaviour on every cross-attention
        (lib/models/architecture/ip_adapter/ip_adapter.py:85-110, :154-160 `set_scale`).  The to_k_ip / to_v_ip weights arrive
        through load_state_dict under their diffusers names `<block>.attn2.processor.to_{k,v}_ip.weight`."""
        self._ip = (int(num_tokens), float(scale))

    def _set_attention(self, cak, B, H, W):
        """Translate the reference's cross_attention_kwargs (mode / ref_dict / is_cfg_guidance) into engine state.  The
        reference fills ref_dict with one tensor per attention layer; here ref_dict receives ONE entry, the K/V store."""
        cak = cak or {}
        mode, ref_dict = cak.get('mode'), cak.get('ref_dict')
        skip = 1 if cak.get('is_cfg_guidance') else 0
        store, ref_mode, rH, rW = None, 0, 0, 0
        if mode is not None and ref_dict is not None:
            if mode == 'w':
                nbytes = _lib.raw('mve_unet_ref_store_bytes')(self._h, B, H, W, skip)
                store = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
                ref_dict[self._REF_KEY] = (store, B, H, W, skip)
                ref_mode = 1
            elif mode in ('r', 'm'):
                store, Bs, rH, rW, sk = ref_dict.pop(self._REF_KEY) if mode == 'r' else ref_dict[self._REF_KEY]
                assert Bs == B and sk == skip, 'reference pass and read pass must have the same batch layout'
                ref_mode = 2
            else:
                raise AssertionError(mode)
        self._ref_keep = store
        _lib.call('mve_unet_set_attention', self._h, self._ip[0], self._ip[1], ref_mode, rH, rW, skip, _lib.ptr(store),
                  store.numel() if store is not None else 0)

    @property
    def weight_bytes(self):
        return _lib.raw('mve_unet_weight_bytes')(self._h)

    # ------------------------------------------------------------------ planning
    def plan(self, B, H, W, ctx_len=77, num_cross_attn_imgs=1, has_residuals=False, io_dtype=None, residuals_nhwc=False):
        ws = ctypes.c_size_t()
        n_ops = ctypes.c_int()
        flops = (ctypes.c_double * 5)()
        _lib.call('mve_unet_plan', self._h, B, H, W, ctx_len, num_cross_attn_imgs, int(has_residuals),
                  _dt(io_dtype or self.dtype), int(residuals_nhwc), ctypes.byref(ws), ctypes.byref(n_ops), flops)
        return dict(workspace_bytes=ws.value, n_ops=n_ops.value, flops=dict(zip(OP_CLASSES, list(flops))))

    def op_table(self):
        """[(phase, class, flops, label)] of the currently cached plan."""
        out, i = [], 0
        cls, fl, lab = ctypes.c_int(), ctypes.c_double(), ctypes.create_string_buffer(96)
        while True:
            ph = _lib.raw('mve_unet_op_info')(self._h, i, ctypes.byref(cls), ctypes.byref(fl), lab, 96)
            if ph < 0:
                break
            out.append((ph, OP_CLASSES[cls.value], fl.value, lab.value.decode()))
            i += 1
        return out

    # ------------------------------------------------------------------ execution
    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def _run(self, phase, sample, timestep, ctx, n_img, down_res, mid_res, out, profile=False, workspace=None,
             residuals_nhwc=False, shape=None):
        B, _, H, W = shape if shape is not None else sample.shape
        io_dtype = ctx.dtype
        assert io_dtype in (torch.float32, torch.float16, torch.bfloat16)
        ctx = ctx.to(self.device).contiguous()
        assert ctx.shape[0] == B and ctx.shape[2] == self.cfg['cross_attention_dim'], ctx.shape
        if sample is not None:
            sample = sample.to(device=self.device, dtype=io_dtype).contiguous()
        t = torch.as_tensor(timestep, dtype=torch.float32, device=self.device).reshape(-1)
        t = t.expand(B).contiguous() if t.numel() == 1 else t.contiguous()
        assert t.numel() == B
        has_res = down_res is not None and mid_res is not None
        keep = []
        res_arr = None
        if has_res:
            want = len(self.cfg['block_out_channels']) * (self.cfg['layers_per_block'] + 1)
            assert len(down_res) == want, f'expected {want} down-block residuals, got {len(down_res)}'
            # residuals produced by mvedit_amd.controlnet are logical NCHW over channels-last storage in the engine dtype:
            # hand the storage over as NHWC without a copy
            cl = lambda r: r.dim() == 4 and r.dtype == self.dtype and r.is_cuda and r.permute(0, 2, 3, 1).is_contiguous()
            if not residuals_nhwc and all(cl(r) for r in down_res) and cl(mid_res):
                residuals_nhwc = True
                down_res = [r.permute(0, 2, 3, 1) for r in down_res]
                mid_res = mid_res.permute(0, 2, 3, 1)
            rdt = self.dtype if residuals_nhwc else io_dtype
            keep = [r.to(device=self.device, dtype=rdt).contiguous() for r in down_res]
            mid_res = mid_res.to(device=self.device, dtype=rdt).contiguous()
            res_arr = (ctypes.c_void_p * want)(*[r.data_ptr() for r in keep])
        info = self.plan(B, H, W, ctx.shape[1], n_img, has_res, io_dtype, residuals_nhwc)
        ws = workspace if workspace is not None else self._workspace(info['workspace_bytes'])
        if out is None and phase != 1:
            out = torch.empty(B, self.cfg['out_channels'], H, W, dtype=io_dtype, device=self.device)
        op_ms = (ctypes.c_float * info['n_ops'])() if profile else None
        with torch.cuda.device(self.device):
            _lib.call('mve_unet_forward', self._h, phase, _lib.ptr(sample), _dt(io_dtype), _lib.ptr(t), _lib.ptr(ctx), B, H, W,
                      ctx.shape[1], n_img, res_arr, _lib.ptr(mid_res) if has_res else None, int(residuals_nhwc),
                      _lib.ptr(out), _lib.ptr(ws), ws.numel(), op_ms, _lib.stream_ptr(self.device))
        if profile:
            return out, list(op_ms)
        return out

    def __call__(self, sample, timestep, encoder_hidden_states, cross_attention_kwargs=None,
                 down_block_additional_residuals=None, mid_block_additional_residual=None, added_cond_kwargs=None,
                 return_dict=False, out=None, **unused):
        """diffusers UNet2DConditionModel.forward signature subset used by the reference."""
        if added_cond_kwargs:
            raise NotImplementedError('added_cond_kwargs (SDXL micro-conditioning) has no reference implementation '
                                      'in MVEdit (SURVEY.md F9)')
        n_img = int((cross_attention_kwargs or {}).get('num_cross_attn_imgs', 1))
        self._set_attention(cross_attention_kwargs, sample.shape[0], sample.shape[2], sample.shape[3])
        res = self._run(0, sample, timestep, encoder_hidden_states, n_img, down_block_additional_residuals,
                        mid_block_additional_residual, out)
        if return_dict:
            return SimpleNamespace(sample=res)
        return (res,)

    forward = __call__

    def profile(self, sample, timestep, encoder_hidden_states, num_cross_attn_imgs=1):
        """-> (out, [(class, label, flops, ms)]) with HIP-event timing around every launch."""
        self._set_attention(None, *[sample.shape[i] for i in (0, 2, 3)])
        out, ms = self._run(0, sample, timestep, encoder_hidden_states, num_cross_attn_imgs, None, None, None, profile=True)
        return out, [(c, lab, fl, m) for (ph, c, fl, lab), m in zip(self.op_table(), ms)]

    # 2-pass mode ---------------------------------------------------------------------------------------
    def enc(self, sample, timestep, encoder_hidden_states, cross_attention_kwargs=None, workspace=None):
        """unet_enc: runs conv_in + down blocks; the returned state handle owns the workspace holding
        (emb, down_block_res_samples, sample) for a later dec()."""
        n_img = int((cross_attention_kwargs or {}).get('num_cross_attn_imgs', 1))
        B, _, H, W = sample.shape
        self._set_attention(cross_attention_kwargs, B, H, W)
        # residual-carrying decode needs the same plan: decided at dec() time, so enc always plans with residual slots
        info = self.plan(B, H, W, encoder_hidden_states.shape[1], n_img, True, encoder_hidden_states.dtype)
        ws = workspace if workspace is not None else torch.empty(info['workspace_bytes'], dtype=torch.uint8, device=self.device)
        st = SimpleNamespace(ws=ws, shape=tuple(sample.shape), timestep=timestep, n_img=n_img)
        self._enc_dec(1, st, sample, encoder_hidden_states, None, None)
        return st

    def dec(self, state, encoder_hidden_states, down_block_additional_residuals=None, mid_block_additional_residual=None,
            cross_attention_kwargs=None):
        self._set_attention(cross_attention_kwargs, state.shape[0], state.shape[2], state.shape[3])
        return self._enc_dec(2, state, None, encoder_hidden_states, down_block_additional_residuals,
                             mid_block_additional_residual)

    def _enc_dec(self, phase, st, sample, ctx, down_res, mid_res):
        B, _, H, W = st.shape
        if phase == 2 and (down_res is None or mid_res is None):
            # unet_dec without ControlNet: feed zero residuals so that the plan (and the enc state layout) is unchanged
            ch = self.cfg['block_out_channels']
            shapes = [(ch[0], H, W)]
            h, w = H, W
            for i, c in enumerate(ch):
                shapes += [(c, h, w)] * self.cfg['layers_per_block']
                if i + 1 < len(ch):
                    h, w = h // 2, w // 2
                    shapes.append((c, h, w))
            down_res = [torch.zeros(B, *s, dtype=ctx.dtype, device=self.device) for s in shapes]
            mid_res = torch.zeros(B, ch[-1], h, w, dtype=ctx.dtype, device=self.device)
        if phase == 1:
            # plan with residual slots, but phase 1 never touches them: pass dummies
            dummy = torch.zeros(8, dtype=ctx.dtype, device=self.device)
            down_res = [dummy] * (len(self.cfg['block_out_channels']) * (self.cfg['layers_per_block'] + 1))
            mid_res = dummy
            return self._run_raw(1, st, sample, ctx, down_res, mid_res)
        return self._run_raw(2, st, None, ctx, down_res, mid_res)

    def _run_raw(self, phase, st, sample, ctx, down_res, mid_res):
        if phase == 1:
            keep_checks = [r for r in down_res]   # dummies: bypass shape validation in _run
            B, _, H, W = st.shape
            io = ctx.dtype
            ctx_c = ctx.to(self.device).contiguous()
            sample = sample.to(device=self.device, dtype=io).contiguous()
            t = torch.as_tensor(st.timestep, dtype=torch.float32, device=self.device).reshape(-1)
            t = t.expand(B).contiguous() if t.numel() == 1 else t.contiguous()
            res_arr = (ctypes.c_void_p * len(keep_checks))(*[r.data_ptr() for r in keep_checks])
            with torch.cuda.device(self.device):
                _lib.call('mve_unet_forward', self._h, 1, _lib.ptr(sample), _dt(io), _lib.ptr(t), _lib.ptr(ctx_c), B, H, W,
                          ctx_c.shape[1], st.n_img, res_arr, _lib.ptr(mid_res), 0, None, _lib.ptr(st.ws), st.ws.numel(), None,
                          _lib.stream_ptr(self.device))
            return None
        return self._run(2, None, st.timestep, ctx, st.n_img, down_res, mid_res, None, workspace=st.ws, shape=st.shape)


def unet_enc(unet, sample, timestep, encoder_hidden_states, cross_attention_kwargs=None, added_cond_kwargs=None):
    """lib/models/architecture/diffusers.py:57-99.  Returns (emb, down_block_res_samples, sample) in the reference;
    here the three live inside one opaque state object, returned in the same 3-tuple positions for drop-in use."""
    st = unet.enc(sample, timestep, encoder_hidden_states, cross_attention_kwargs)
    return st, st, st


def unet_dec(unet, emb, down_block_res_samples, sample, encoder_hidden_states, cross_attention_kwargs=None,
             down_block_additional_residuals=None, mid_block_additional_residual=None):
    """lib/models/architecture/diffusers.py:102-164."""
    return unet.dec(emb, encoder_hidden_states, down_block_additional_residuals, mid_block_additional_residual,
                    cross_attention_kwargs)
