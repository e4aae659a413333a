"""Mirror of the hot-path methods of the reference's `lib.pipelines.adapter3d_mixin.Adapter3DMixin`.

`get_noise_pred` keeps the reference's name, argument order/meaning and result
(lib/pipelines/adapter3d_mixin.py:68-135): per chunk ControlNets -> UNet -> drop the reference half, then
the classifier-free-guidance combine.  `self.unet` is a `mvedit_amd.unet.UNet2DConditionEngine` (or any object
with the diffusers call signature); `self.controlnet` is whatever the runner loaded (diffusers
MultiControlNetModel in the reference) or None.

MI355X-first difference: the reference walks 2*ceil(V/diff_bs) chunks of diff_bs<=6 views because of
24 GB-class GPUs.  With 288 GB of HBM the chunks are concatenated and the UNet runs ONCE over all 2V
images (`fuse_chunks=True`, the default): weights are streamed once per step instead of once per chunk and
every GEMM sees M = 2V*H*W rows.  The kernels are batch-invariant (a row's reduction order does not depend
on the batch), so the result is bitwise identical to the chunked walk (tests/test_pipeline_mixin.py).
"""
from copy import copy

import torch

from .. import ops
from ..unet import unet_dec, unet_enc


class Adapter3DMixin:
    fuse_chunks = True

    def get_tgt_masks(self, tgt_images, seg_padding):
        """lib/pipelines/adapter3d_mixin.py:14-19: foreground masks of the denoised views ([1, V, H, W, 3] -> [1, V, H, W, 1]) from
        `self.segmentation` (mvedit_amd.segmentor.TracerUniversalB7Engine) with the pipeline's background colour override."""
        from .utils import do_segmentation
        tgt_images = tgt_images.squeeze(0).clip(min=0, max=1).permute(0, 3, 1, 2)
        images_masked = do_segmentation(tgt_images, self.segmentation, padding=seg_padding, bg_color=self.bg_color)
        return images_masked[:, 3][None, ..., None]

    def _unet_chunk(self, latent, prompt_embeds, ctrl_images, ctrl_depths, extra_control, t, tile_weight, depth_weight,
                    added_cond_kwargs):
        """One chunk of the reference loop body (adapter3d_mixin.py:85-128)."""
        latent_shape = latent.size()
        latent_size = latent_shape[3]
        paired = latent_shape[2] == 2 * latent_shape[3]          # [b, 4, 2H, W]: reference image stacked on the view
        if paired:
            cross_attention_kwargs = dict(num_cross_attn_imgs=2)
            unet_in = latent.reshape(*latent_shape[:2], 2, latent_shape[3], latent_shape[3]).permute(0, 2, 1, 3, 4) \
                .reshape(latent_shape[0] * 2, latent_shape[1], latent_shape[3], latent_shape[3])
            cn_in = latent[:, :, -latent_size:]
            unet_embeds = prompt_embeds.unsqueeze(1).expand(-1, 2, -1, -1).reshape(-1, *prompt_embeds.shape[1:])
            cn_embeds = prompt_embeds
        else:
            cross_attention_kwargs = None
            unet_in = cn_in = latent
            unet_embeds = cn_embeds = prompt_embeds
        down_res = mid_res = None
        if getattr(self, 'controlnet', None) is not None:
            down_res, mid_res = self.controlnet(
                cn_in, t, encoder_hidden_states=cn_embeds,
                controlnet_cond=[ctrl_images, ctrl_depths] + list(extra_control),
                conditioning_scale=[tile_weight, depth_weight] + [1.0] * len(extra_control),
                guess_mode=False, added_cond_kwargs=added_cond_kwargs, return_dict=False)
            if paired:   # zero residuals for the reference rows (adapter3d_mixin.py:110-116)
                down_res = [torch.stack([torch.zeros_like(r), r], dim=1).reshape(-1, *r.shape[1:]) for r in down_res]
                mid_res = torch.stack([torch.zeros_like(mid_res), mid_res], dim=1).reshape(-1, *mid_res.shape[1:])
        out = self.unet(unet_in, t, encoder_hidden_states=unet_embeds, cross_attention_kwargs=cross_attention_kwargs,
                        down_block_additional_residuals=down_res, mid_block_additional_residual=mid_res,
                        added_cond_kwargs=added_cond_kwargs, return_dict=False)[0]
        if paired:
            out = out.view(latent_shape[0], 2, latent_shape[1], latent_shape[3], latent_shape[3])[:, 1]
        return out

    def get_noise_pred(self, latent_batches, prompt_embeds_batches, ctrl_images_batches, ctrl_depths_batches,
                       t, tile_weight, depth_weight, guidance_scale, extra_control_batches=None,
                       added_cond_kwargs_batches=None, adapter_scale=None):
        if ctrl_depths_batches is None:
            ctrl_depths_batches = [None] * len(latent_batches)
        if extra_control_batches is None:
            extra_control_batches = []
        same_shape = len({tuple(b.shape[1:]) for b in latent_batches}) == 1
        fuse = self.fuse_chunks and same_shape and added_cond_kwargs_batches is None and len(latent_batches) > 1
        if fuse:
            cat = lambda bs: None if bs[0] is None else torch.cat(list(bs), dim=0)
            cond = self._cat_shared_cond if self._controlnet_shares_cond() else cat
            noise_pred = self._unet_chunk(
                cat(latent_batches), cat(prompt_embeds_batches), cond(ctrl_images_batches), cond(ctrl_depths_batches),
                [cond(e) for e in extra_control_batches], t, tile_weight, depth_weight, None)
        else:
            outs = []
            for i, (lat, emb, ci, cd, *extra) in enumerate(zip(latent_batches, prompt_embeds_batches, ctrl_images_batches,
                                                                ctrl_depths_batches, *extra_control_batches)):
                ack = None if added_cond_kwargs_batches is None else {k: v[i] for k, v in added_cond_kwargs_batches.items()}
                outs.append(self._unet_chunk(lat, emb, ci, cd, extra, t, tile_weight, depth_weight, ack))
            noise_pred = torch.cat(outs, dim=0)
        noise_pred_uncond, noise_pred_text = noise_pred.chunk(2)
        if adapter_scale is not None:
            return adapter_scale * (noise_pred_text - noise_pred_uncond)
        if noise_pred.is_cuda:
            return ops.cfg_combine(noise_pred_uncond, noise_pred_text, guidance_scale).to(noise_pred.dtype)
        return guidance_scale * noise_pred_text + (1 - guidance_scale) * noise_pred_uncond

    # ------------------------------------------------------------------------------------------------------------------
    # 2-pass mode (adapter3d_mixin.py:137-317): pass 1 runs encoder + decoder without the tile ControlNet and keeps the
    # encoder state; pass 2 evaluates the tile (and depth) ControlNet on the fresh renders and re-runs the DECODER only.
    # ------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _pair(latent, prompt_embeds):
        """[b,4,2H,W] (reference image stacked on the view) -> UNet batch of 2b images; adapter3d_mixin.py:156-167."""
        shp = latent.size()
        if shp[2] != 2 * shp[3]:
            return False, None, latent, latent, prompt_embeds, prompt_embeds
        unet_in = latent.reshape(*shp[:2], 2, shp[3], shp[3]).permute(0, 2, 1, 3, 4).reshape(shp[0] * 2, shp[1], shp[3], shp[3])
        embeds = prompt_embeds.unsqueeze(1).expand(-1, 2, -1, -1).reshape(-1, *prompt_embeds.shape[1:])
        return True, dict(num_cross_attn_imgs=2), unet_in, latent[:, :, -shp[3]:], embeds, prompt_embeds

    @staticmethod
    def _pad_pair(down_res, mid_res):
        """zero residuals for the reference rows (adapter3d_mixin.py:186-192)."""
        down_res = [torch.stack([torch.zeros_like(r), r], dim=1).reshape(-1, *r.shape[1:]) for r in down_res]
        mid_res = torch.stack([torch.zeros_like(mid_res), mid_res], dim=1).reshape(-1, *mid_res.shape[1:])
        return down_res, mid_res

    def _controlnet_shares_cond(self):
        """True when every ControlNet of the runner accepts fewer conditioning images than batch items (the native engines do)."""
        cn = getattr(self, 'controlnet', None)
        nets = getattr(cn, 'nets', [cn] if cn is not None else [])
        return len(nets) > 0 and all(getattr(n, 'shares_cond', False) for n in nets)

    detect_repeated_cond = True      # one device comparison + host read per distinct control tensor (cached, see _cat_shared_cond); False: object identity only

    @staticmethod
    def _as_one_tensor(bs):
        """Chunks that are consecutive views of ONE storage -- what `x.split(diff_bs)` returns -- as that tensor again, without a copy; else None."""
        b0 = bs[0]
        if b0.dim() == 0 or not b0.is_contiguous():
            return None
        st, off, total = b0.untyped_storage().data_ptr(), b0.storage_offset(), 0
        for b in bs:
            if (b.dtype != b0.dtype or b.shape[1:] != b0.shape[1:] or not b.is_contiguous() or b.untyped_storage().data_ptr() != st
                    or b.storage_offset() != off):
                return None
            off += b.numel()
            total += b.shape[0]
        # canonical contiguous strides from the shape, NOT b0.stride(): torch calls a tensor contiguous whatever the stride of a size-1 leading
        # dimension, so a one-item chunk may carry any stride(0)
        strides, acc = [], 1
        for d in reversed(b0.shape[1:]):
            strides.append(acc)
            acc *= int(d)
        strides = (acc,) + tuple(reversed(strides))
        return torch.as_strided(b0, (total,) + tuple(b0.shape[1:]), strides, b0.storage_offset())

    def _cat_shared_cond(self, bs):
        """Fused conditioning batch of the ControlNets.  The reference builds the CFG halves of the control images from the same data, in two forms:
        `ctrl_images.split(diff_bs) * 2` (mvedit_3d_pipeline.py:1232, :1417 -- the use_reference branch and the 2-pass methods): the second half of
        the list IS the first half (object identity, nothing is read); and `torch.cat([ctrl_images] * 2).split(diff_bs)` (:1238-1241, the ordinary
        1-pass branch): the chunks are consecutive views of one tensor whose halves hold the same values -- recognised by viewing the chunks as
        that tensor again (no copy) and comparing its halves on the device (`torch.equal`: ~50 MB read + one host read per call of a >= 100 ms
        step; `detect_repeated_cond = False` turns the comparison off).  Either way only ONE half is handed over, and the engines run the
        conditioning embedding once for both halves of the batch (ControlNetEngine.run: item b uses image b mod len(cond); bit-identical to
        repeating the images).  Anything else is concatenated whole."""
        if bs[0] is None:
            return None
        bs = list(bs)
        n = len(bs) // 2
        if len(bs) % 2 == 0 and n > 0 and all(bs[i] is bs[i + n] for i in range(n)):
            return torch.cat(bs[:n], dim=0)
        if self.detect_repeated_cond and len(bs) > 1:
            full = self._as_one_tensor(bs)
            if full is not None and full.shape[0] >= 2 and full.shape[0] % 2 == 0:
                h = full.shape[0] // 2
                # the verdict is cached per (storage, version, geometry): control images are constant over a denoise loop, so the device
                # comparison and its host read happen once per loop, not once per step (the read stalled launch-ahead)
                key = (full.untyped_storage().data_ptr(), full._version, full.storage_offset(), tuple(full.shape), full.dtype)
                cache = self.__dict__.setdefault('_repeated_cond_cache', {})
                same = cache.get(key)
                if same is None:
                    if len(cache) > 64:
                        cache.clear()
                    same = cache[key] = bool(torch.equal(full[:h], full[h:]))
                return full[:h] if same else full
        return torch.cat(bs, dim=0)

    def _sub_controlnet(self, nets):
        """MultiControlNetModel(self.controlnet.nets[a:b]) of the reference, without importing diffusers here."""
        return type(self.controlnet)(nets)

    def _maybe_fuse(self, *batch_lists, cond=()):
        """Concatenate the diff_bs chunks into one batch when shapes allow (see the module docstring).  `cond`: positions of the lists that are
        ControlNet conditioning images -- those go through _cat_shared_cond when the ControlNets accept shared images."""
        lat = batch_lists[0]
        if not (self.fuse_chunks and len(lat) > 1 and len({tuple(b.shape[1:]) for b in lat}) == 1):
            return batch_lists
        shared = self._controlnet_shares_cond()
        cat = lambda bs, c: None if bs is None else ([None] if bs[0] is None else [self._cat_shared_cond(bs) if (c and shared) else torch.cat(list(bs), dim=0)])
        return tuple(cat(b, i in cond) for i, b in enumerate(batch_lists))

    def get_noise_pred_p1(self, latent_batches, prompt_embeds_batches, t, guidance_scale, ctrl_depths_batches=None,
                          depth_weight=None, extra_control_batches=None, cond_noisy_latent_batches=None,
                          added_cond_kwargs_batches=None):
        extra_control_batches = extra_control_batches or []
        if added_cond_kwargs_batches is None:
            fused = self._maybe_fuse(latent_batches, prompt_embeds_batches, ctrl_depths_batches, cond_noisy_latent_batches,
                                     *extra_control_batches, cond=(2,) + tuple(range(4, 4 + len(extra_control_batches))))
            latent_batches, prompt_embeds_batches, ctrl_depths_batches, cond_noisy_latent_batches = fused[:4]
            extra_control_batches = list(fused[4:])
        n = len(latent_batches)
        ctrl_depths_batches = ctrl_depths_batches or [None] * n
        noise_pred, dec_args, dec_kwargs = [], [], []
        for i in range(n):
            latent, embeds, depths = latent_batches[i], prompt_embeds_batches[i], ctrl_depths_batches[i]
            extra = [e[i] for e in extra_control_batches]
            paired, cak, unet_in, cn_in, unet_embeds, cn_embeds = self._pair(latent, embeds)
            ack = None if added_cond_kwargs_batches is None else {k: v[i] for k, v in added_cond_kwargs_batches.items()}
            skip = 2 if depths is None else 1                       # nets[0] = tile (pass 2), nets[1] = depth
            down_res = mid_res = None
            nets = getattr(getattr(self, 'controlnet', None), 'nets', [])
            if len(nets) > skip:
                down_res, mid_res = self._sub_controlnet(nets[skip:])(
                    cn_in, t, encoder_hidden_states=cn_embeds,
                    controlnet_cond=extra if depths is None else [depths] + extra,
                    conditioning_scale=[1.0] * len(extra) if depths is None else [depth_weight] + [1.0] * len(extra),
                    guess_mode=False, added_cond_kwargs=ack, return_dict=False)
                if paired:
                    down_res, mid_res = self._pad_pair(down_res, mid_res)
            enc_cak, dec_cak = cak, cak
            if cond_noisy_latent_batches is not None:             # reference attention: write pass over the condition latents
                assert cak is None
                ref_enc, ref_dec = dict(), dict()
                cond_state = unet_enc(self.unet, cond_noisy_latent_batches[i], t, encoder_hidden_states=unet_embeds,
                                      added_cond_kwargs=ack, cross_attention_kwargs=dict(mode='w', ref_dict=ref_enc))
                unet_dec(self.unet, *cond_state, encoder_hidden_states=unet_embeds,
                         cross_attention_kwargs=dict(mode='w', ref_dict=ref_dec))
                enc_cak, dec_cak = dict(mode='r', ref_dict=ref_enc), dict(mode='m', ref_dict=ref_dec)
            state = unet_enc(self.unet, unet_in, t, encoder_hidden_states=unet_embeds, cross_attention_kwargs=enc_cak,
                             added_cond_kwargs=ack)
            dec_args.append(state)
            dec_kwargs.append(dict(encoder_hidden_states=unet_embeds, cross_attention_kwargs=dec_cak,
                                   down_block_additional_residuals=down_res, mid_block_additional_residual=mid_res))
            out = unet_dec(self.unet, *dec_args[-1], **dec_kwargs[-1])
            if paired:
                out = out.view(latent.shape[0], 2, latent.shape[1], latent.shape[3], latent.shape[3])[:, 1]
            noise_pred.append(out)
        noise_pred = torch.cat(noise_pred, dim=0)
        uncond, text = noise_pred.chunk(2)
        return self._cfg(uncond, text, guidance_scale), dec_args, dec_kwargs

    def get_noise_pred_p2(self, latent_batches, prompt_embeds_batches, dec_args, dec_kwargs, t, guidance_scale,
                          ctrl_images_batches, tile_weight, ctrl_depths_batches=None, depth_weight=None,
                          added_cond_kwargs_batches=None, guess_mode=False, adapter_scale=None, ctrl_text_embedding=True):
        if len(dec_args) == 1 and len(latent_batches) > 1:       # pass 1 fused the chunks: fuse the same way
            latent_batches, prompt_embeds_batches, ctrl_images_batches, ctrl_depths_batches = self._maybe_fuse(
                latent_batches, prompt_embeds_batches, ctrl_images_batches, ctrl_depths_batches, cond=(2, 3))
        n = len(latent_batches)
        ctrl_depths_batches = ctrl_depths_batches or [None] * n
        noise_pred = []
        for i in range(n):
            latent, embeds, depths = latent_batches[i], prompt_embeds_batches[i], ctrl_depths_batches[i]
            paired = latent.shape[2] == 2 * latent.shape[3]
            cn_in = latent[:, :, -latent.shape[3]:] if paired else latent
            ack = None if added_cond_kwargs_batches is None else {k: v[i] for k, v in added_cond_kwargs_batches.items()}
            if ctrl_text_embedding:
                cn_embeds, cn_ack = embeds, ack
            else:                                                  # adapter3d_mixin.py:268-277
                cn_embeds = self.negative_prompt_embeds.to(cn_in).expand(latent.shape[0], -1, -1)
                neg = getattr(self, 'negative_added_cond_kwargs', None)
                cn_ack = None if neg is None else {k: v.to(cn_in).expand(latent.shape[0], *[-1] * (v.dim() - 1)) for k, v in neg.items()}
            down_res, mid_res = self._sub_controlnet(self.controlnet.nets[:1 if depths is None else 2])(
                cn_in, t, encoder_hidden_states=cn_embeds,
                controlnet_cond=[ctrl_images_batches[i]] if depths is None else [ctrl_images_batches[i], depths],
                conditioning_scale=[tile_weight] if depths is None else [tile_weight, depth_weight],
                guess_mode=guess_mode, added_cond_kwargs=cn_ack, return_dict=False)
            if paired:
                down_res, mid_res = self._pad_pair(down_res, mid_res)
            kw = copy(dec_kwargs[i])
            if kw['down_block_additional_residuals'] is not None:
                down_res = [a + b for a, b in zip(down_res, kw['down_block_additional_residuals'])]
            if kw['mid_block_additional_residual'] is not None:
                mid_res = mid_res + kw['mid_block_additional_residual']
            kw.update(down_block_additional_residuals=down_res, mid_block_additional_residual=mid_res)
            out = unet_dec(self.unet, *dec_args[i], **kw)
            if paired:
                out = out.view(latent.shape[0], 2, latent.shape[1], latent.shape[3], latent.shape[3])[:, 1]
            noise_pred.append(out)
        noise_pred = torch.cat(noise_pred, dim=0)
        uncond, cond = noise_pred.chunk(2)
        if adapter_scale is not None:
            return adapter_scale * (cond - uncond)
        return self._cfg(uncond, cond, guidance_scale)

    @staticmethod
    def _cfg(uncond, text, guidance_scale):
        if uncond.is_cuda:
            return ops.cfg_combine(uncond, text, guidance_scale).to(uncond.dtype)
        return guidance_scale * text + (1 - guidance_scale) * uncond
