"""mvedit_amd -- MI355X (gfx950) engine for the denoise -> render -> reconstruct hot path of MVEdit.

Host-side modules mirror the reference's operator interfaces for that path only:
    mvedit_amd.raymarching        <-> lib.ops.raymarching
    mvedit_amd.unet               <-> the diffusers UNet seam + lib.models.architecture.diffusers.unet_enc/unet_dec
    mvedit_amd.pipelines          <-> lib.pipelines.adapter3d_mixin.Adapter3DMixin.get_noise_pred*
    mvedit_amd.parallel           view partitioning + the one all-gather of the multi-GPU design
Arithmetic lives in mvedit_amd/csrc (hand-written HIP) behind the C ABI of include/mvedit_amd.h.
"""
__version__ = '0.1.0'
