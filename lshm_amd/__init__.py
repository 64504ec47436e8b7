"""lshm_amd - MI355X-native (gfx950) implementation of the LSHM cascaded-autoencoder +
k-harmonic-means training step, behind the reference's own module API.

    from lshm_amd.lofar_models import AutoEncoderCNN2, AutoEncoder1DCNN, Kmeans   # drop-in modules
    from lshm_amd.kharmonic_lofar import KHarmonicTrainer, TrainConfig            # fused step engine
"""
from . import _lib
from .lofar_models import AutoEncoder1DCNN, AutoEncoderCNN, AutoEncoderCNN2, Kmeans
from .kharmonic_lofar import KHarmonicTrainer, TrainConfig
from .functional import augmented_loss, fft_features
from .fft_cascade import fft_cascade_forward, fft_cascade_latents
from .lbfgsnew import LBFGSNew

__all__ = ["AutoEncoderCNN2", "AutoEncoderCNN", "AutoEncoder1DCNN", "Kmeans", "KHarmonicTrainer",
           "TrainConfig", "augmented_loss", "fft_features", "fft_cascade_forward", "fft_cascade_latents",
           "LBFGSNew"]
__version__ = "0.1.0"
