"""MI355X-native latent score-model sampling path (drop-in for the reference's module-level API).

Sub-modules mirror the reference's flat files for the hot path only:

    unet_model        Unet                       (reference unet_model.py:189-323)
    unet_openai       UNetModel                  (reference unet_openai.py:361-575)
    h_vae_model_copy  ResVAE/ResAE/ResVAEN/ResAEN.decoder   (reference h_vae_model_copy.py)
    h_vae_model       CelebAAttrNewBN[AE].decoder            (reference h_vae_model.py:712-899)
    sde_helper2       VPSDE/subVPSDE/VESDE, em_predictor, corrector, uncond_sampler

All numerics run in hand-written HIP kernels for gfx950 behind the C ABI declared in
``include/sbmae.h`` (library ``csrc/libsbmae_hip.so``).  There is no CPU fallback: any compute
entry point raises if the library or a GPU is missing.
"""
__version__ = "0.1.0"
