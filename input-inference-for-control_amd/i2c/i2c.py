"""Same import path as the reference: `from i2c.i2c import I2cGraph`."""
from .exp_types import CubatureQuadrature, GaussHermiteQuadrature, Linearize  # noqa: F401
from .graph import CHECK_COVAR, DEBUG_PLOTS, PLOT_TIKZ, I2cCell, I2cGraph  # noqa: F401
