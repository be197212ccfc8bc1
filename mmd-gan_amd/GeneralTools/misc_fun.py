"""FLAGS: the attribute bag every reference module reads (reference misc_fun.py:25-60).

Scripts mutate it before importing the rest (my_test_cifar.py:2-5), so it is a module-level
singleton here too.  Two attributes are additions of this build: SYNTHETIC_DATA (no dataset in
the container: uniform [-1,1] batches generated on the device) and NUM_GPUS (alias of the
reference's dormant `num_gpus`, misc_fun.py:28).
"""

_DEFAULTS = dict(
    num_gpus=1, EPSI=1e-10, SILENT_MODE=False,
    # the reference records the stack it ran on (misc_fun.py:33-36); this build's stack instead:
    BACKEND='hip-gfx950', ROCM_VERSION='7.2',
    DEFAULT_IN='MMD-GAN/Data/', DEFAULT_OUT='MMD-GAN/Results/', DEFAULT_DOWNLOAD='MMD-GAN/Data/',
    INCEPTION_V1=None, INCEPTION_V3=None, PLT_ACC=None, PLT_KEY=None,
    IMAGE_FORMAT='channels_first', IMAGE_FORMAT_ALIAS='NCHW',
    WEIGHT_INITIALIZER='default',          # 'default' | 'sn_paper' | 'pg_paper'
    SPECTRAL_NORM_MODE='default',          # 'default' (= PICO) | 'sn_paper' (PIM: flattened-kernel power iteration)
    SYNTHETIC_DATA=False,
)


class SetFlag(object):
    def __init__(self):
        for key, value in _DEFAULTS.items():
            setattr(self, key, value)

    @property
    def NUM_GPUS(self):
        return self.num_gpus

    @NUM_GPUS.setter
    def NUM_GPUS(self, value):
        self.num_gpus = int(value)

    def print(self, info, force_print=False):
        if force_print or not self.SILENT_MODE:
            print(info)


FLAGS = SetFlag()
