// TEMPORARY: ResNet-path entry points not yet implemented return CCST_EINVAL (replaced by resnet_ops.hip).
#include "../../include/ccst_hip.h"
void ccst_set_error(const char* fmt, ...);
#define NI(name) { ccst_set_error(#name ": not implemented yet"); return CCST_EINVAL; }
extern "C" {
int ccst_conv2d_bwd_weight_f32(const CcstConvDesc*, const float*, const float*, float*, int, void*, int64_t, void*) NI(bwd_weight)
int ccst_bn_train_fwd_f32(const float*, const float*, const float*, float*, float*, float, float, const float*, int, float*, float*, float*, int64_t, int, void*, int64_t, void*) NI(bn_train_fwd)
int ccst_bn_eval_fwd_f32(const float*, const float*, const float*, const float*, const float*, float, const float*, int, float*, int64_t, int, void*) NI(bn_eval_fwd)
int ccst_bn_train_bwd_f32(const float*, const float*, const float*, const float*, const float*, const float*, int, float*, float*, float*, float*, int64_t, int, void*, int64_t, void*) NI(bn_train_bwd)
int64_t ccst_bn_workspace_bytes(int64_t, int) { return 0; }
int ccst_maxpool3s2_fwd_f32(const float*, float*, int, int, int, int, int, int, void*) NI(maxpool_fwd)
int ccst_maxpool3s2_bwd_f32(const float*, const float*, float*, int, int, int, int, int, int, void*) NI(maxpool_bwd)
int ccst_avgpool_fwd_f32(const float*, float*, int, int, int, void*) NI(avgpool_fwd)
int ccst_avgpool_bwd_f32(const float*, float*, int, int, int, void*) NI(avgpool_bwd)
int ccst_linear_fwd_f32(const float*, const float*, const float*, float*, int, int, int, void*) NI(linear_fwd)
int ccst_linear_bwd_f32(const float*, const float*, const float*, float*, float*, float*, int, int, int, void*) NI(linear_bwd)
int ccst_softmax_ce_f32(const float*, const int64_t*, float*, float*, int32_t*, int, int, void*) NI(softmax_ce)
int ccst_sgd_f32(float*, const float*, float, int64_t, void*) NI(sgd)
int ccst_scale_f32(float*, float, int64_t, void*) NI(scale)
}
