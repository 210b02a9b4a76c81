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

# the reference splits each model into a definition mixin (<Name>Def) and the known model built on it (env_def.py:233-298, ...);
# here one class carries both, under both names. The learned-model and Furuta definitions are outside this build (known models only).
PendulumDef = PendulumKnown
BaseCartpoleDef = CartpoleDef = CartpoleKnown
DoubleCartpoleDef = DoubleCartpoleKnown
