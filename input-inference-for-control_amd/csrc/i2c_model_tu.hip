// One translation unit per (model, dtype): compiled with
//   -DI2C_TU_MODEL=<struct in i2c_models.hpp> -DI2C_TU_REAL=<double|float> -DI2C_TU_OPS=<ops_<name>_<f64|f32|f64s>>
//   [-DI2C_TU_STORE=float]   storage type of the per-cell buffers (default: I2C_TU_REAL); f64s = double arithmetic, float storage
// (see build.py). All kernels of the pair are instantiated here and nowhere else.
//   [-DI2C_TU_HEADER="<path>"]  an OUT-OF-TREE model: the header that defines struct I2C_TU_MODEL in namespace i2c (derived from
//                               ModelDefaults, i2c_models.hpp); `python build.py --model <path>`
#include "i2c_impl.hpp"
#ifdef I2C_TU_HEADER
#include I2C_TU_HEADER
#endif

#if !defined(I2C_TU_MODEL) || !defined(I2C_TU_REAL) || !defined(I2C_TU_OPS)
#error "compile with -DI2C_TU_MODEL=... -DI2C_TU_REAL=... -DI2C_TU_OPS=..."
#endif

namespace i2c {
#ifndef I2C_TU_STORE
#define I2C_TU_STORE I2C_TU_REAL
#endif
const ModelOps* I2C_TU_OPS() { return make_ops<I2C_TU_MODEL, I2C_TU_REAL, I2C_TU_STORE>(); }
}  // namespace i2c
