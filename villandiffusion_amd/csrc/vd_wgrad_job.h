// Device job table of the grouped weight-gradient launches (shared by vd_gemm.hip and vd_presplit.hip; not part of the C ABI).
#pragma once
#include "vd_common.h"

// Device job table of the GROUPED launch (vd_conv_wgrad_group_*): several weight gradients of one kernel class in ONE grid, so that
// the ~768 resident workgroups are shared by all of them -- each job is then split over far fewer workgroups (long K ranges, few
// slabs) than when it has to fill the chip alone.
struct vd_wgrad_job {
    vd_wgrad_desc d;           // d.ws = this job's slab region (or null when it is not split)
    int32_t first_block;       // compute grid: blocks [first_block, first_block + gx * gy); multiple of 8 (XCD-aware remap inside a job)
    int32_t gx, gy;            // tiles, splits
    int32_t ks_per;            // K-steps per split
    int32_t first_rblock;      // reduce grid: blocks [first_rblock, first_rblock + rblocks) (0 blocks when gy == 1)
    int32_t rblocks;
    int32_t pad_[2];
};

__device__ __forceinline__ int wgrad_find_job(const vd_wgrad_job* __restrict__ jobs, int n, int blk, bool reduce) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {                                            // last job whose first block <= blk (block-uniform)
        const int mid = (lo + hi + 1) >> 1;
        if ((reduce ? jobs[mid].first_rblock : jobs[mid].first_block) <= blk) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

