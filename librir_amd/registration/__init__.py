from .masked_registration_ecc import MaskedRegistratorECC, find_transform_ecc_translation, manage_computation_and_tries  # noqa: F401
from .device_registration import DeviceRegistratorECC  # noqa: F401
