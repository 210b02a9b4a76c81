"""Drop-in mirror of the reference's ``i2c`` package for the cubature hot path."""
