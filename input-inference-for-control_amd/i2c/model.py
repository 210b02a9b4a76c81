"""Same import path as the reference's i2c/model.py: `from i2c.model import make_env_model`."""
from .known_models import (  # noqa: F401
    ENVIRONMENTS,
    CartpoleKnown,
    DoubleCartpoleKnown,
    KnownModel,
    LinearExact,
    LinearMinimumEnergy,
    PendulumKnown,
    PendulumKnownActReg,
    PlanarQuadrotor,
    make_env_model,
)

BaseModelKnown = KnownModel
BaseModel = KnownModel  # (model.py:19-78: the common base; the learned-model classes are outside this build)
LinearBase = LinearExact
