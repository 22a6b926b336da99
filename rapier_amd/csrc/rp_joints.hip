// rp_joints.hip — joint colouring / layout kernels and the MULTI-mode joint launches.
//
//   * colouring: ParallelInteractionGroups::group_interactions (solver/interaction_groups.rs:59-197) is
//     a serial greedy pass over the joints in edge order that keeps a joint's stored colour while it
//     is free of the bodies' contact colours and of the colours taken by earlier joints.  While every
//     stored colour is still free the pass is the identity (staged_joint_colors_still_free,
//     staged_island_solver/joints.rs:462-480), which a parallel check establishes each step; otherwise
//     the greedy order is reproduced exactly as a wavefront over the joints' dependency DAG inside one
//     workgroup (a joint decides once the previous joint, by edge index, at either of its bodies has).
//   * layout: colours with >= 64 joints (JOINT_BATCH * LAYOUT_REF_WORKERS / 2, joints.rs:352) are
//     parallel stages in ascending colour order, everything else is the serial overflow, colour-major
//     in edge order (single_group_joint_layout, joints.rs:331-460).
#include "rp_joints.h"

#define RP_JOINT_PARALLEL_MIN 64

RP_DEV unsigned long long jld_u64(unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RP_DEV unsigned jld_u32(unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Does any joint's stored colour collide with its bodies' contact colours (or is it uncoloured)?
__global__ void k_joint_color_check(DevWorld w) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= w.n_joints) return;
    if (!joint_live(w, j)) return; // removed or sleeping joint: not in this step's selection
    int color = w.j_color[j];
    int b1 = w.j_b1[j], b2 = w.j_b2[j];
    bool bad = color >= 128;
    if (!bad) {
        unsigned bit = 1u << (color & 31);
        if (b1 >= 0 && (w.b_cmask[4 * b1 + (color >> 5)] & bit)) bad = true;
        if (b2 >= 0 && (w.b_cmask[4 * b2 + (color >> 5)] & bit)) bad = true;
    }
    if (bad) w.flags[FL_JOINT_DIRTY] = 1;
}

// The greedy pass as a wavefront over its dependency DAG (the scheme of k_color_pairs, rp_narrowphase.hip): a joint's decision
// depends on the joints with smaller edge indices at its (at most two) bodies only, so every body's joints form a chain in edge
// order, a joint is ready when its predecessor at either body has decided, and a thread that decides a joint goes straight on
// to the successor it releases.  (The bidding rounds this replaces rescanned every joint per round: 21 ms for the first step of
// b3d_joint_grid.)
RP_DEV int jld_i32(int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RP_DEV void joint_color_body(DevWorld &w) { // one workgroup
    if (!w.flags[FL_JOINT_DIRTY]) return;
    const int nj = w.n_joints, nb = w.n_bodies;
    const int tid = threadIdx.x, nt = blockDim.x;
    __shared__ int cursor, n_cur, n_next;
    if (tid == 0) { cursor = 0; n_cur = 0; n_next = 0; }
    for (int b = tid; b < nb; b += nt) { for (int q = 0; q < 4; ++q) w.bj_cmask[4 * b + q] = 0; }
    __threadfence(); __syncthreads();
    // per-body lists of the live joints (joints outside the selection keep their stored colour and take no part)
    for (int j = tid; j < nj; j += nt) {
        const bool live = joint_live(w, j);
        w.j_tmp[j] = live ? 1 : 0;
        if (!live) continue;
        int b1 = w.j_b1[j], b2 = w.j_b2[j], first = 0;
        if (b1 >= 0 && atomicAdd(&w.col_cnt[b1], 1) == 0) first |= 1;
        if (b2 >= 0 && b2 != b1 && atomicAdd(&w.col_cnt[b2], 1) == 0) first |= 2;
        w.jc_first[j] = first;
    }
    __threadfence(); __syncthreads();
    for (int j = tid; j < nj; j += nt) {
        if (!w.j_tmp[j]) continue;
        int b1 = w.j_b1[j], b2 = w.j_b2[j], first = w.jc_first[j];
        if (first & 1) w.col_begin[b1] = atomicAdd(&cursor, jld_i32(&w.col_cnt[b1]));
        if (first & 2) w.col_begin[b2] = atomicAdd(&cursor, jld_i32(&w.col_cnt[b2]));
    }
    __threadfence(); __syncthreads();
    for (int j = tid; j < nj; j += nt) {
        if (!w.j_tmp[j]) continue;
        int b1 = w.j_b1[j], b2 = w.j_b2[j];
        if (b1 >= 0) w.jc_list[jld_i32(&w.col_begin[b1]) + atomicAdd(&w.col_fill[b1], 1)] = j;
        if (b2 >= 0 && b2 != b1) w.jc_list[jld_i32(&w.col_begin[b2]) + atomicAdd(&w.col_fill[b2], 1)] = j;
    }
    __threadfence(); __syncthreads();
    for (int j = tid; j < nj; j += nt) { // rank by edge index inside each body's list
        if (!w.j_tmp[j]) continue;
        int b1 = w.j_b1[j], b2 = w.j_b2[j];
        int2 rk = make_int2(-1, -1);
        for (int side = 0; side < 2; ++side) {
            int b = side ? b2 : b1;
            if (b < 0 || (side && b2 == b1)) continue;
            int beg = jld_i32(&w.col_begin[b]), n = jld_i32(&w.col_cnt[b]), q = 0;
            for (int k = 0; k < n; ++k) q += jld_i32(&w.jc_list[beg + k]) < j;
            w.jc_sorted[beg + q] = j;
            if (side) rk.y = q; else rk.x = q;
        }
        w.jc_rank[j] = rk;
        w.jc_deps[j] = (rk.x > 0) + (rk.y > 0);
    }
    __threadfence(); __syncthreads();
    for (int j = tid; j < nj; j += nt) {
        if (!w.j_tmp[j]) continue;
        int b1 = w.j_b1[j], b2 = w.j_b2[j];
        int2 rk = w.jc_rank[j];
        int s1 = -1, s2 = -1;
        if (rk.x >= 0 && rk.x + 1 < jld_i32(&w.col_cnt[b1])) s1 = jld_i32(&w.jc_sorted[jld_i32(&w.col_begin[b1]) + rk.x + 1]);
        if (rk.y >= 0 && rk.y + 1 < jld_i32(&w.col_cnt[b2])) s2 = jld_i32(&w.jc_sorted[jld_i32(&w.col_begin[b2]) + rk.y + 1]);
        w.jc_succ[j] = make_int2(s1, s2);
        if (rk.x <= 0 && rk.y <= 0) w.jc_q[atomicAdd(&n_cur, 1)] = j;
    }
    __threadfence(); __syncthreads();
    for (int j = tid; j < nj; j += nt) { // the shared per-body counters go back to rest (k_color_pairs uses them too)
        if (!w.j_tmp[j]) continue;
        int b1 = w.j_b1[j], b2 = w.j_b2[j];
        if (b1 >= 0) { w.col_cnt[b1] = 0; w.col_fill[b1] = 0; }
        if (b2 >= 0) { w.col_cnt[b2] = 0; w.col_fill[b2] = 0; }
    }
    int *qc = w.jc_q, *qn = w.jc_q + (nj > 0 ? nj : 1);
    for (;;) {
        const int n = n_cur;
        __syncthreads();
        if (n == 0) break;
        for (int f = tid; f < n; f += nt) {
            int j = jld_i32(&qc[f]);
            for (int hops = 0; j >= 0; ++hops) {
                // (bounded: a thread that walked a long chain to its end would hold the round open while every other ready joint waits)
                if (hops == 8) { qn[atomicAdd(&n_next, 1)] = j; break; }
                const int b1 = w.j_b1[j], b2 = w.j_b2[j];
                const int2 su = w.jc_succ[j];
                unsigned m[4] = {0, 0, 0, 0};
                if (b1 >= 0) for (int q = 0; q < 4; ++q) m[q] |= jld_u32(&w.bj_cmask[4 * b1 + q]) | jld_u32(&w.b_cmask[4 * b1 + q]);
                if (b2 >= 0) for (int q = 0; q < 4; ++q) m[q] |= jld_u32(&w.bj_cmask[4 * b2 + q]) | jld_u32(&w.b_cmask[4 * b2 + q]);
                int stored = w.j_color[j], color = 128;
                if (stored < 128 && !((m[stored >> 5] >> (stored & 31)) & 1u)) color = stored;      // keep_or_pick
                else if (b1 >= 0 && b2 >= 0) { for (int c = 0; c < RP_DYNAMIC_COLOR_COUNT; ++c) if (!((m[c >> 5] >> (c & 31)) & 1u)) { color = c; break; } }
                else { for (int c = 127; c >= 0; --c) if (!((m[c >> 5] >> (c & 31)) & 1u)) { color = c; break; } }
                if (color < 128) {
                    unsigned bit = 1u << (color & 31);
                    if (b1 >= 0) atomicOr(&w.bj_cmask[4 * b1 + (color >> 5)], bit);
                    if (b2 >= 0) atomicOr(&w.bj_cmask[4 * b2 + (color >> 5)], bit);
                }
                w.j_color[j] = color;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the mask bits have reached L2 before a successor can be released (one workgroup, L2 atomics only: see k_color_pairs)
                int next = -1;
                if (su.x >= 0 && atomicSub(&w.jc_deps[su.x], 1) == 1) next = su.x;
                if (su.y >= 0 && atomicSub(&w.jc_deps[su.y], 1) == 1) { if (next < 0) next = su.y; else qn[atomicAdd(&n_next, 1)] = su.y; }
                j = next;
            }
        }
        __threadfence(); __syncthreads();
        if (tid == 0) { n_cur = n_next; n_next = 0; }
        int *tmp = qc; qc = qn; qn = tmp;
        __syncthreads();
    }
}

// Stage layout of the coloured joints (one workgroup).
RP_DEV void joint_layout_body(DevWorld &w) { // one workgroup
    if (!w.flags[FL_JOINT_DIRTY]) return;
    const int nj = w.n_joints;
    __shared__ int count[RP_NUM_COLORS], begin[RP_NUM_COLORS], cursor[RP_NUM_COLORS], stage_of[RP_NUM_COLORS];
    __shared__ int n_ovf, ovf_begin;
    for (int c = threadIdx.x; c < RP_NUM_COLORS; c += blockDim.x) count[c] = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < nj; j += blockDim.x) if (joint_live(w, j)) atomicAdd(&count[w.j_color[j]], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int pos = 0, nst = 0;
        for (int c = 0; c < RP_NUM_COLORS; ++c) {
            stage_of[c] = -1;
            if (c < 128 && count[c] >= RP_JOINT_PARALLEL_MIN) {
                w.j_stage_begin[nst] = pos; w.j_stage_count[nst] = count[c];
                begin[c] = pos; cursor[c] = pos; stage_of[c] = nst; pos += count[c]; nst++;
            }
        }
        ovf_begin = pos; n_ovf = 0;
        int live = 0; for (int c = 0; c < RP_NUM_COLORS; ++c) live += count[c];
        w.flags[FL_NJ_STAGES] = nst; w.flags[FL_NJ_OVF_BEGIN] = pos; w.flags[FL_NJ_OVF_COUNT] = live - pos;
        // the serial tail is colour-major too, and a colour below 128 is body-disjoint whatever its size: the tile sweeps (rp_tiles.hip)
        // take the small colours as further stages behind the parallel ones.  [RP_NUM_COLORS] of the two arrays: how many stages that
        // makes in all, and how many joints sit in the one colour that is NOT disjoint (128: such a world does not tile)
        int nall = nst, p2 = pos;
        for (int c = 0; c < 128; ++c) if (count[c] > 0 && count[c] < RP_JOINT_PARALLEL_MIN) { w.j_stage_begin[nall] = p2; w.j_stage_count[nall] = count[c]; p2 += count[c]; nall++; }
        w.j_stage_count[RP_NUM_COLORS] = nall; w.j_stage_begin[RP_NUM_COLORS] = count[128];
    }
    __syncthreads();
    // parallel colours: order inside a colour is free (body-disjoint); overflow: collected, then ranked
    for (int j = threadIdx.x; j < nj; j += blockDim.x) {
        if (!joint_live(w, j)) continue;
        int c = w.j_color[j];
        if (stage_of[c] >= 0) w.j_order[atomicAdd(&cursor[c], 1)] = j;
        else w.j_tmp[atomicAdd(&n_ovf, 1)] = j;
    }
    __threadfence(); __syncthreads();
    const int no = n_ovf, ob = ovf_begin;
    for (int i = threadIdx.x; i < no; i += blockDim.x) { // colour-major, edge order within a colour
        int j = w.j_tmp[i];
        long long key = ((long long)w.j_color[j] << 32) | j;
        int rank = 0;
        for (int q = 0; q < no; ++q) { int jq = w.j_tmp[q]; long long kq = ((long long)w.j_color[jq] << 32) | jq; rank += kq < key; }
        w.j_order[ob + rank] = j;
    }
    __threadfence(); __syncthreads();
    if (threadIdx.x == 0) { w.flags[FL_JOINT_DIRTY] = 0; w.flags[FL_FLOW_DIRTY] = 1; } // the joint sweep order changed: re-rank (rp_flow.hip)
}

// colouring + stage layout of the joints in ONE single-workgroup launch (both are gated by FL_JOINT_DIRTY: a clean step pays one
// early exit instead of two)
__global__ void __launch_bounds__(1024) k_joint_color_layout(DevWorld w) {
    if (!w.flags[FL_JOINT_DIRTY]) return; // (uniform: raised by k_joint_color_check or the host before this launch)
    joint_color_body(w);
    __threadfence(); __syncthreads();
    joint_layout_body(w);
}

// ---- MULTI mode launches -----------------------------------------------------------------------
__global__ void k_joint_update(DevWorld w, int substep_id) {
    int stride = gridDim.x * blockDim.x;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < w.n_joints; j += stride) if (joint_live(w, j)) joint_update_one(w, j, substep_id);
}
__global__ void __launch_bounds__(256) k_joint_stage(DevWorld w, int stage, int wo_bias, int warmstart) {
    if (stage >= w.flags[FL_NJ_STAGES]) return;
    int beg = w.j_stage_begin[stage], cnt = w.j_stage_count[stage];
    int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) joint_solve_one(w, w.j_order[beg + i], wo_bias != 0, warmstart != 0);
}
// joint stages the host did not launch + the serial overflow, in one workgroup
__global__ void __launch_bounds__(1024) k_joint_tail(DevWorld w, int first, int wo_bias, int warmstart) {
    int nst = w.flags[FL_NJ_STAGES];
    joint_tail_sweep(w, first < nst ? first : nst, wo_bias != 0, warmstart != 0);
}
__global__ void k_joint_writeback(DevWorld w) {
    int stride = gridDim.x * blockDim.x;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < w.n_joints; j += stride) if (joint_live(w, j)) joint_writeback_one(w, j);
}

static int joint_blocks(const DevWorld &w) { int b = (w.n_joints + 255) / 256; if (b > 2048) b = 2048; return b < 1 ? 1 : b; }

// after the contact colouring of the narrow phase (the contact masks are final for this step)
void rp_launch_joint_coloring(const DevWorld &w, hipStream_t st) {
    if (w.n_joints == 0) return;
    hipLaunchKernelGGL(k_joint_color_check, dim3((w.n_joints + 255) / 256), dim3(256), 0, st, w);
    hipLaunchKernelGGL(k_joint_color_layout, dim3(1), dim3(1024), 0, st, w);
}
void rp_launch_joint_update(const DevWorld &w, hipStream_t st, int substep_id) {
    if (w.n_joints == 0) return;
    hipLaunchKernelGGL(k_joint_update, dim3(joint_blocks(w)), dim3(256), 0, st, w, substep_id);
}
void rp_launch_joint_sweep(const DevWorld &w, hipStream_t st, int parallel_stages, int wo_bias, int warmstart) {
    if (w.n_joints == 0) return;
    int blocks = joint_blocks(w);
    for (int s = 0; s < parallel_stages; ++s) hipLaunchKernelGGL(k_joint_stage, dim3(blocks), dim3(256), 0, st, w, s, wo_bias, warmstart);
    hipLaunchKernelGGL(k_joint_tail, dim3(1), dim3(1024), 0, st, w, parallel_stages, wo_bias, warmstart);
}
void rp_launch_joint_writeback(const DevWorld &w, hipStream_t st) {
    if (w.n_joints == 0) return;
    hipLaunchKernelGGL(k_joint_writeback, dim3(joint_blocks(w)), dim3(256), 0, st, w);
}
