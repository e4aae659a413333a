"""mvedit_amd -- MI355X (gfx950) engine for the denoise -> render -> reconstruct hot path of MVEdit.

Host-side modules mirror the reference's operator interfaces for that path only:
    mvedit_amd.raymarching        <-> lib.ops.raymarching
    mvedit_amd.unet               <-> the diffusers UNet seam + lib.models.architecture.diffusers.unet_enc/unet_dec
    mvedit_amd.controlnet / vae / image_enhancer <-> the diffusers ControlNet / AutoencoderKL seams, SRVGGNetCompact
    mvedit_amd.pipelines          <-> lib.pipelines.adapter3d_mixin.Adapter3DMixin.get_noise_pred*, lib.core.diffusion, the outer loop's
                                      schedules / camera pruning / ray sampling / highpass (lib.pipelines.utils, mvedit_3d_pipeline)
    mvedit_amd.nerf               <-> BaseNeRF.render, VolumeRenderer, iNGPDecoder.point_decode (+ backward, Adam)
    mvedit_amd.mesh_ops           <-> MeshRenderer, DMTet, Mesh.auto_normal, edge_dilation, the mesh regularisers
    mvedit_amd.recon_loss / lpips <-> the image-space losses of nerf_optim / mesh_optim, LPIPSLoss
    mvedit_amd.tonemapping / shencoder <-> Tonemapping, lib.ops.shencoder
    mvedit_amd.parallel           view partitioning, the one all-gather of the multi-GPU design, the scene re-sync
Arithmetic lives in mvedit_amd/csrc (hand-written HIP) behind the C ABI of include/mvedit_amd.h.
"""
__version__ = '0.1.0'
