// Entry points of an OUT-OF-TREE model library (lib/libi2c_model_<name>.so, built by `python build.py --model <header>`): the
// three per-dtype tables of the model's translation units (i2c_model_tu.hip compiled with -DI2C_TU_HEADER) behind plain C symbols
// that i2c_load_model() of the main library resolves with dlsym. The reference's counterpart is "any object with dim_*, forward,
// observe, observe_terminal_x is a model" (i2c/model.py:19-44, 154-156; i2c/env_def.py:34-82): here a model is a header with a
// functor struct, compiled without touching the tree.
//   -DI2C_PLUGIN_NAME=<identifier>   the ops tables are i2c::ops_<identifier>_{f64,f32,f64s}
#include "i2c_entry.hpp"

#ifndef I2C_PLUGIN_NAME
#error "compile with -DI2C_PLUGIN_NAME=<identifier>"
#endif
#define I2C_PASTE3_(a, b, c) a##b##c
#define I2C_PASTE3(a, b, c) I2C_PASTE3_(a, b, c)
#define I2C_STR_(x) #x
#define I2C_STR(x) I2C_STR_(x)

namespace i2c {
const ModelOps* I2C_PASTE3(ops_, I2C_PLUGIN_NAME, _f64)();
const ModelOps* I2C_PASTE3(ops_, I2C_PLUGIN_NAME, _f32)();
const ModelOps* I2C_PASTE3(ops_, I2C_PLUGIN_NAME, _f64s)();
}  // namespace i2c

extern "C" {
__attribute__((visibility("default"))) int i2c_model_abi_version(void) { return I2C_ABI_VERSION; }
__attribute__((visibility("default"))) const char* i2c_model_name(void) { return I2C_STR(I2C_PLUGIN_NAME); }
__attribute__((visibility("default"))) const I2cModelOps* i2c_model_ops(int dtype) {
  const i2c::ModelOps* ops = dtype == I2C_F64   ? i2c::I2C_PASTE3(ops_, I2C_PLUGIN_NAME, _f64)()
                             : dtype == I2C_F32 ? i2c::I2C_PASTE3(ops_, I2C_PLUGIN_NAME, _f32)()
                             : dtype == I2C_F64_F32S ? i2c::I2C_PASTE3(ops_, I2C_PLUGIN_NAME, _f64s)()
                                                     : nullptr;
  return reinterpret_cast<const I2cModelOps*>(ops);
}
}
