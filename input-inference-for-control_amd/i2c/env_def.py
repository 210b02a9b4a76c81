"""Same import path as the reference's i2c/env_def.py (known-model definitions only)."""
from .known_models import (  # noqa: F401
    CartpoleKnown,
    DoubleCartpoleKnown,
    KnownModel as BaseDef,
    LinearExact as LinearDef,
    LinearMinimumEnergy as LinearMinimumEnergyDef,
    PendulumKnown,
    PendulumKnownActReg,
)
