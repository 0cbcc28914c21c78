// Launchers of csrc/net.hip that the bf16 training executor (csrc/train_bf16.hip) shares with the fp32 one.  Not part of the C-ABI.
#pragma once
#include "common.h"
#include "layout.h"

struct LinrShortRange { int64_t b, e; int rows; };     // parameters [b, e) hold partials in the first `rows` slab rows only
__attribute__((visibility("hidden")))
int linr_bwd_tail_launch(const linr_frame* f, const Layout& L, const float* P, const float* gx0, const float* hid, float* big,
                         float* gsum, int nb, const LinrShortRange* sh, int nsh, hipStream_t stream);
__attribute__((visibility("hidden")))
int linr_adam_step_launch(const Layout& L, float* params, const float* gsum, float* exp_avg, float* exp_avg_sq, double lr, int64_t step,
                          const int64_t* scale_steps_h, double beta1, double beta2, double eps, double weight_decay, hipStream_t s);
__attribute__((visibility("hidden"))) int linr_wg_blocks_for(int64_t rows);
// live kernel timing / poison hook of csrc/net.hip (include/linr_hip.h: linr_prof_*) around a launch of another file
struct LinrProf {
    void* impl;
    LinrProf(hipStream_t s, int kind, int passes);
    ~LinrProf();
    LinrProf(const LinrProf&) = delete;
    LinrProf& operator=(const LinrProf&) = delete;
};
