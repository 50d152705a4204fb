/* Laboratory surface of libgom_hip.so: entry points and modes that were built, measured and NOT adopted (DESIGN.md / LABBOOK.md say why).
 * They exist only in a library compiled with -DGOM_LAB (python scripts/exp_build.py lab -DGOM_LAB); the default build neither compiles
 * nor exports them, and include/gom_hip.h -- the product ABI -- does not declare them.
 *   - gom_state_set_frame_optimizer: the Adam launch as the last launch of the frame step's recorded graph (measured slower than a plain
 *     launch behind the graph: 13.96 k against 14.09 k frames/s);
 *   - GOM_OPT_BWD_MODE 2: the (sub-range, 4 x 4 block) render backward, one item per DPP row (csrc/lab/seg_bwd_blk.hpp: 200 us against 152);
 *   - GOM_OPT_BWD_MODE 3: the records render backward (csrc/lab/rec_bwd.hpp + the REC = true instantiations of k_seg_T / k_seg_fwd): the forward leaves
 *     one record per blending (pixel, entry) pair, the backward walks a lane per entry over its records instead of replaying the list (set it
 *     BEFORE the forward; 298 us + 55 us of forward overhead against the replay's 153: profiles/r04_records_backward.txt).  Its record buffers
 *     (24 bytes x 8 per unit of pair capacity) are allocated for a state in that mode only. */
#ifndef GOM_HIP_LAB_H
#define GOM_HIP_LAB_H
#include "gom_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Attach that step to a state's frame step: gom_frame_forward_backward / gom_batch_forward_backward then end with the Adam launch
 * (`grads` = the flat buffer the frame's g_* pointers are views of; the device step counter is required), so that forward, backward AND the
 * optimizer are one recorded graph (a plain launch behind a graph launch starts ~9 us late).  Not applied by GOM_FRAME_FORWARD_ONLY calls.
 * n == 0 or params == NULL detaches it. */
int gom_state_set_frame_optimizer(GomState *s, int64_t n, float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int32_t n_segments,
                                  const int64_t *seg_begin, const float *seg_lr, int64_t *step_device, float lr_decay_steps, float beta1, float beta2,
                                  float eps, float grad_scale);


#ifdef __cplusplus
}
#endif
#endif /* GOM_HIP_LAB_H */
