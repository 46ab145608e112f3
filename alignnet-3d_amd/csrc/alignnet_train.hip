// Training-side C ABI (include/alignnet_hip.h).  Filled in by the training milestone; until
// then each entry point reports "not implemented" rather than silently doing nothing.
#include "engine.h"

static int nyi(alignnet_handle* h, const char* what)
{
  if (h) h->err = std::string(what) + ": not implemented in this build";
  return 2;
}

extern "C" int alignnet_eval_loss(alignnet_handle* h, const alignnet_labels*, int32_t, float*, float*) { return nyi(h, "alignnet_eval_loss"); }
extern "C" int alignnet_train_step(alignnet_handle* h, const float*, const float*, const alignnet_labels*, int32_t, const float*,
                                   alignnet_step_result*, const alignnet_outputs*) { return nyi(h, "alignnet_train_step"); }
extern "C" int alignnet_train_forward_backward(alignnet_handle* h, const float*, const float*, const alignnet_labels*, int32_t,
                                               const float*, alignnet_step_result*, const alignnet_outputs*) { return nyi(h, "alignnet_train_forward_backward"); }
extern "C" int alignnet_grad_buffer(alignnet_handle* h, float**, size_t*) { return nyi(h, "alignnet_grad_buffer"); }
extern "C" int alignnet_apply_gradients(alignnet_handle* h, float) { return nyi(h, "alignnet_apply_gradients"); }
extern "C" int alignnet_get_grad(alignnet_handle* h, const char*, float*, size_t) { return nyi(h, "alignnet_get_grad"); }
extern "C" int alignnet_comm_unique_id(uint8_t*) { return 2; }
extern "C" int alignnet_comm_init(alignnet_handle* h, int32_t, int32_t, const uint8_t*) { return nyi(h, "alignnet_comm_init"); }
extern "C" int alignnet_comm_allreduce_grads(alignnet_handle* h) { return nyi(h, "alignnet_comm_allreduce_grads"); }
extern "C" int alignnet_save(alignnet_handle* h, const char*) { return nyi(h, "alignnet_save"); }
extern "C" int alignnet_load(alignnet_handle* h, const char*, int32_t) { return nyi(h, "alignnet_load"); }
