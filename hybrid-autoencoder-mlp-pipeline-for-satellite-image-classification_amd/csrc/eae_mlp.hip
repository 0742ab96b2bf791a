// external MLP kernels -- TEMPORARY stubs so the library exports the full ABI while the kernels are being written
#include "eae_internal.h"
#define NI return eae_set_error(EAE_ERR_STATE, "MLP engine not implemented yet")
extern "C" int eae_mlp_layout(int, int, long long*, long long*) { NI; }
extern "C" int eae_mlp_create(int, int, int, eae_mlp**) { NI; }
extern "C" int eae_mlp_destroy(eae_mlp*) { return 0; }
extern "C" int eae_mlp_bind(eae_mlp*, float*, float*, float*, float*, float*, long long*) { NI; }
extern "C" int eae_mlp_set_adam_step(eae_mlp*, long long) { NI; }
extern "C" int eae_mlp_forward(eae_mlp*, void*, const float*, int, int, unsigned long long, const float*, float*) { NI; }
extern "C" int eae_mlp_train_step(eae_mlp*, void*, const float*, const long long*, int, float, float, unsigned long long, const float*, float*, float*) { NI; }
extern "C" int eae_mlp_eval_step(eae_mlp*, void*, const float*, const long long*, int, float*, float*) { NI; }
