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
import torch

from .. import ops


class Adapter3DMixin:
    fuse_chunks = True

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
                down_res = [torch.stack([torch.zeros_like(r), r], dim=1).view(-1, *r.shape[1:]) for r in down_res]
                mid_res = torch.stack([torch.zeros_like(mid_res), mid_res], dim=1).view(-1, *mid_res.shape[1:])
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
            noise_pred = self._unet_chunk(
                cat(latent_batches), cat(prompt_embeds_batches), cat(ctrl_images_batches), cat(ctrl_depths_batches),
                [cat(e) for e in extra_control_batches], t, tile_weight, depth_weight, None)
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
