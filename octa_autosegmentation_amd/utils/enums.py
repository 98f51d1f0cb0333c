"""String-valued phase / task enums of the configs (reference utils/enums.py:19-26): members compare equal to, hash like
and print as their YAML spelling, so `config[Phase.TRAIN]` and `config["Train"]` are the same lookup."""
from enum import Enum


class _StrEnum(Enum):
    def __eq__(self, other):
        if isinstance(other, _StrEnum):
            return type(other) is type(self) and self.value == other.value
        if isinstance(other, str):
            return self.value == other
        return NotImplemented

    def __hash__(self):
        return hash(self.value)

    def __str__(self):
        return str(self.value)

    def __repr__(self):
        return repr(self.value)


class Phase(_StrEnum):
    TRAIN = "Train"
    VALIDATION = "Validation"
    TEST = "Test"


class Task(_StrEnum):
    VESSEL_SEGMENTATION = "ves-seg"
    GAN_VESSEL_SEGMENTATION = "gan-ves-seg"
