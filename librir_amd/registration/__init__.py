from .masked_registration_ecc import MaskedRegistratorECC, find_transform_ecc_translation  # noqa: F401
from .device_registration import DeviceRegistratorECC  # noqa: F401
