// Token path engine (combined embedding, encoder, decoders, focal loss) -- C ABI entry points.
#include "../../include/tcvn_hip.h"
#include "head_plan.h"

extern "C" {
int tcvn_head_create(const tcvn_head_cfg* cfg, tcvn_head** out) { (void)cfg; (void)out; return -100; }
void tcvn_head_destroy(tcvn_head* p) { (void)p; }
int tcvn_head_num_slots(const tcvn_head* p) { (void)p; return 0; }
int tcvn_head_slot(const tcvn_head* p, int i, char* name, int cap, int64_t* numel, int* kind) { return -100; }
int tcvn_head_bind(tcvn_head* p, void* const* data, void* const* grad) { return -100; }
int64_t tcvn_head_workspace_bytes(const tcvn_head* p, int batch, int max_prongs, int n_prongs) { return 0; }
int tcvn_head_forward(tcvn_head* p, int batch, int max_prongs, int n_prongs, const float* rows, const int32_t* tok_row,
                      float* event_logits, float* prong_logits, void* workspace, int64_t workspace_bytes, int train,
                      uint64_t seed, void* stream) { return -100; }
int tcvn_head_loss(tcvn_head* p, int batch, int max_prongs, const float* event_logits, const float* prong_logits,
                   const int64_t* event_targets, const int8_t* prong_targets, float* losses, float* accs, void* workspace,
                   int64_t workspace_bytes, void* stream) { return -100; }
int tcvn_head_backward(tcvn_head* p, int batch, int max_prongs, int n_prongs, const float* rows, const int32_t* tok_row,
                       float loss_scale, float* d_rows, void* workspace, int64_t workspace_bytes, void* stream) { return -100; }
}
