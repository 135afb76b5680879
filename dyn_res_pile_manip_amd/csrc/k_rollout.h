// km_rollout: a whole H-step rollout (planners.py:302-370) of small piles in ONE launch.
//
// A workgroup of the whole-sample kernels owns its samples from the first step to the last: nothing a sample needs
// at step t + 1 is produced by another workgroup, so the rollout needs no chip-wide synchronisation at all -- only the
// launch boundaries of the step-by-step pipeline (k_graph -> km_prop3, H times) impose one.  Here the workgroup loops
// over the steps itself:
//     per step:  push impulses + displaced positions of its samples -> LDS          (gen_s_delta, planners.py:211-257)
//                neighbour lists, one thread per receiver, two sweeps over the sample (graph_receiver, k_graph.h)
//                particle encoder -> row order -> three propagation steps -> prediction (prop3_step, k_mlp_split.h)
//                (workgroups of at most 256 rows with paired tiles or the kept rows: the lists are built by the waves that have
//                no encoder tile, WHILE the others run the encoder -- one barrier and the shorter of the two phases fewer per
//                step)
// with the node matrices resident in LDS for the whole launch (the step-by-step pipeline refills 98 KB per rollout
// step), no launch gap, no graph launch, and no end-of-launch wait for the slowest workgroup of the chip: a workgroup
// that finishes a step early starts the next one, so the imbalance between workgroups averages out over the H steps
// instead of being paid H times.  Every arithmetic instruction is the step-by-step pipeline's (the same device
// functions, inlined): the two paths produce the same bits (tests/test_gpu_fullsize.py).
//
// Used for samples of up to DRP_ROLLOUT_MAX_N particles (default 64) in workgroups of up to 704 rows
// ( the kernel itself takes any sample whose workgroup holds at most KM_ROLLOUT_MAX_ROWS rows: the displaced positions for the plain sweep sit in the edge chain's LDS
// region between two steps).  Measured against the step-by-step pipeline on one box (tools/ab_rollout.sh, 1024 samples
// x 10 steps): 10 particles 0.575 -> 0.486 ms per MPC iteration, 20: 0.805 -> 0.707, 50: 1.377 -> 1.362, 64: 1.639 ->
// 1.543, 100: 2.84 -> 2.86, 150: 4.12 -> 4.59 (there the x-strip build of k_graph.h beats the in-kernel plain sweep).
// What a rollout step costs inside (tools/rollout_stamps.py, 50 particles, wave 0): lists 10.6 us, encoder phase with
// its two weight swaps 15 us, the three propagation steps 110 us -- one tile per wave, i.e. pure latency.
#pragma once
#include "k_graph.h"
#include "k_mlp_split.h"

#define KM_ROLLOUT_MAX_ROWS 3072        // spw * N: 16 B of LDS per row in the edge chain's 53 KB

// The kernel's arguments live in device memory and are re-read at the top of every rollout step: as kernel arguments
// proper (40 pointers and sizes + the camera) they would sit in scalar registers for the whole launch, and the
// propagation steps' slot loop would run between spills of them.
struct RolloutArgs {
    const uint16_t* sw; const uint16_t* sw6; const float* mw;
    const float* s_in;                       // state before the first step: row b % nb, N * 3 floats each
    float* states;                           // [B][H][N][3]: step t of sample b at (b * H + t) * N * 3
    const float* attr; const float* dens;    // [nb][N], [nb]
    const float* actions;                    // [B][H][4]
    float* s_delta; int16_t* nbr_idx; uint8_t* nbr_cnt;
    float* proj_a; float* proj_b; float* c_node; float* eff;
    const float* cself; const uint8_t* cself_ok;
    float4* ecache; size_t ec_stride;        // ECACHE: the edge-chain cache (prop_tiles), ec_stride float4 per workgroup
    unsigned long long* work;                // WORK: PROP_WORK_* counters
    int N, B, spw, nb, H, order_rows;
    float thr, re_scale, re_inv;
    DrpCam cam;
};

template <bool PAIR, bool ECACHE, bool WORK, bool ONE /* no more tiles than waves per workgroup: rows handed from step to step in registers */>
__global__ void __launch_bounds__(64 * PROP_WAVES)
km_rollout(const RolloutArgs* __restrict__ args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef PROP_STAMPS
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    WorkClock wclk;
    if constexpr (WORK) wclk = work_clock_begin();
    const Prop3Lds P = prop3_lds(lds);
    prop3_fill_resident(P, args->sw, args->sw6, args->mw);   // stays for all H steps; the first barrier of step 0 publishes it
    float4* p4 = reinterpret_cast<float4*>(P.wsp_f);         // displaced positions of the workgroup's rows, between two steps
    const int H = args->H;
#ifdef ROLLOUT_STAMPS
    const bool roll_on = threadIdx.x == 0 && (blockIdx.x & 31) == 0;
    unsigned long long roll_t = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll 1
    for (int t = 0; t < H; ++t) {
        // Nothing derived may stay live across the propagation steps below (their slot loop leaves no vector register
        // to spare and few scalar ones; what the compiler hoists out of this loop as invariant -- per-thread addresses,
        // the sample bases of a dozen buffers -- it spills and reloads around every tile): the thread index and the
        // argument block are re-read behind opaque copies every step.
        // (read through the constant address space: scalar loads, the values land in scalar registers)
        int tid = threadIdx.x;
        typedef const RolloutArgs __attribute__((address_space(4))) * ArgsPtr;
        ArgsPtr a = (ArgsPtr)args;
        asm volatile("" : "+v"(tid), "+s"(a));
        const int N = a->N, B = a->B, spw = a->spw, nbat = a->nb;
        const int b0 = blockIdx.x * spw, nb = min(spw, B - b0);
        const int wg_rows = (nb > 0 ? nb : 0) * N;
        const float inv_N = 1.0f / (float)N;
        const size_t hstride = (size_t)H * N * 3;
        float* states = a->states;
        const float* s_prev = (t == 0) ? a->s_in : states + (size_t)(t - 1) * N * 3;
        const int prev_mod = (t == 0) ? nbat : B;
        const size_t prev_stride = (t == 0) ? (size_t)N * 3 : hstride;
        float* s_delta = a->s_delta;
        int16_t* nbr_idx = a->nbr_idx;
        uint8_t* nbr_cnt = a->nbr_cnt;
        // the previous step's predictions (this workgroup's own stores) and the last readers of the edge chain's LDS
        __syncthreads();
        ROLL_STAMP(0);                               // waiting for the other waves at the end of a rollout step (+ the entry fill)
        DrpCam cam;
        for (int q = 0; q < 12; ++q) cam.m[q] = a->cam.m[q];
        cam.gs = a->cam.gs; cam.fx = a->cam.fx; cam.fy = a->cam.fy; cam.cx = a->cam.cx; cam.cy = a->cam.cy;
        const float* actions = a->actions;
        if constexpr (PAIR || ONE) {
            // Paired tiles (at most 192 rows, the host's rule) or no more unpaired tiles than waves (at most 256 rows): the
            // lists are built by the waves WITHOUT an encoder tile while the others run the encoder (prop3_step<ENC_PRE>; with a
            // tile for every wave, by all of them behind it -- still one barrier fewer).  For that the encoder's matrices are
            // filled in here, and the displaced positions go behind them: the two matrices leave 4 KB of the edge chain's
            // region free, 16 B per row.
            lds_fill(P.wsp_f, reinterpret_cast<const float*>(a->sw6) + S6_PE2 * 4, 2 * 1536 * 4, tid);
            lds_fill(P.pe0_f, reinterpret_cast<const float*>(a->sw6) + S6_PE0 * 4, 384 * 4, tid);
            lds_fill(P.rows_e, a->mw + R_PE2_B, 192, tid);
            if (tid == 0) *P.tile_ctr = PROP_WAVES;
            float4* p4x = reinterpret_cast<float4*>(P.wsp_f + 2 * 1536 * 4);
            static_assert((S_TOTAL * 4 - 2 * 1536 * 4) * sizeof(float) >= 256 * sizeof(float4), "positions of 256 rows behind the encoder's matrices");
            for (int r = tid; r < wg_rows; r += 64 * PROP_WAVES) {
                int m, i;
                divmod_small(r, N, inv_N, m, i);
                const int b = b0 + m;
                const PushFrame f = push_frame(cam, actions + ((size_t)b * H + t) * 4);
                const float* s = s_prev + (size_t)(b % prev_mod) * prev_stride;
                const float x = s[i * 3 + 0], y = s[i * 3 + 1], z = s[i * 3 + 2];
                float ox, oy, oz;
                push_delta(f, x, y, z, ox, oy, oz);
                float* sd = s_delta + ((size_t)b0 * N + r) * 3;
                sd[0] = ox; sd[1] = oy; sd[2] = oz;
                p4x[r] = make_float4(__fadd_rn(x, ox), __fadd_rn(y, oy), __fadd_rn(z, oz), 0.0f);      // gnn_dyn.py:224
            }
            __syncthreads();
            ROLL_STAMP(1);                           // encoder matrices, impulses and displaced positions
            const float thr = a->thr;
            const int self_first = a->cself != nullptr ? 1 : 0;
            auto lists = [&](int w, int nw) {
                for (int r = w * 64 + (tid & 63); r < wg_rows; r += nw * 64) {
                    int m, i;
                    divmod_small(r, N, inv_N, m, i);
                    const size_t row = (size_t)b0 * N + r;
                    graph_receiver(p4x + m * N, N, i, thr, self_first, nbr_idx + row * DRP_K, nbr_cnt + row);
                }
            };
            prop3_step<false, PAIR, true, ECACHE, WORK, ONE, true>(P, a->sw, a->sw6, a->mw, s_prev, prev_mod, prev_stride, a->attr, nbat, a->dens, nbat, nbr_idx, nbr_cnt,
                              a->proj_a, a->proj_b, a->c_node, a->eff, N, B, spw, s_delta, states + (size_t)t * N * 3, hstride,
                              a->cself, a->cself_ok, nullptr, nullptr, a->re_scale, a->re_inv, a->order_rows, tid,
                              ECACHE ? a->ecache + (size_t)blockIdx.x * a->ec_stride : nullptr, a->work PROP_STAMPS_ARG, lists);
        } else {
        {
            for (int r = tid; r < wg_rows; r += 64 * PROP_WAVES) {
                int m, i;
                divmod_small(r, N, inv_N, m, i);
                const int b = b0 + m;
                const PushFrame f = push_frame(cam, actions + ((size_t)b * H + t) * 4);
                const float* s = s_prev + (size_t)(b % prev_mod) * prev_stride;
                const float x = s[i * 3 + 0], y = s[i * 3 + 1], z = s[i * 3 + 2];
                float ox, oy, oz;
                push_delta(f, x, y, z, ox, oy, oz);
                float* sd = s_delta + ((size_t)b0 * N + r) * 3;
                sd[0] = ox; sd[1] = oy; sd[2] = oz;
                p4[r] = make_float4(__fadd_rn(x, ox), __fadd_rn(y, oy), __fadd_rn(z, oz), 0.0f);      // gnn_dyn.py:224
            }
        }
        __syncthreads();
        ROLL_STAMP(1);                               // impulses and displaced positions
        {
            const float thr = a->thr;
            const int self_first = a->cself != nullptr ? 1 : 0;
            for (int r = tid; r < wg_rows; r += 64 * PROP_WAVES) {
                int m, i;
                divmod_small(r, N, inv_N, m, i);
                const size_t row = (size_t)b0 * N + r;
                graph_receiver(p4 + m * N, N, i, thr, self_first, nbr_idx + row * DRP_K, nbr_cnt + row);
            }
        }
        __syncthreads();                             // the lists are written, the positions no longer needed
        ROLL_STAMP(2);                               // neighbour lists
        prop3_step<false, PAIR, true, ECACHE, WORK, ONE>(P, a->sw, a->sw6, a->mw, s_prev, prev_mod, prev_stride, a->attr, nbat, a->dens, nbat, nbr_idx, nbr_cnt,
                          a->proj_a, a->proj_b, a->c_node, a->eff, N, B, spw, s_delta, states + (size_t)t * N * 3, hstride,
                          a->cself, a->cself_ok, nullptr, nullptr, a->re_scale, a->re_inv, a->order_rows, tid,
                          ECACHE ? a->ecache + (size_t)blockIdx.x * a->ec_stride : nullptr, a->work PROP_STAMPS_ARG);
        }
#ifdef ROLLOUT_STAMPS
        roll_t = __builtin_amdgcn_s_memrealtime();   // prop3_step keeps its own clock
        if (roll_on) atomicAdd(&g_roll_stamps[15], 1ull);
#endif
    }
    if constexpr (WORK) work_clock_end(wclk, args->work);
}
#define KM_ROLLOUT_LDS KM_PROP3_LDS

// ---- a handful of samples with impulses that are DATA (the trainer's forward pass, predict_one_step): the neighbour lists
// (k_graph_q4: positions + impulses -> lists) and the particle encoder (km_node_encode_split: impulses, attributes, densities ->
// effects, node constants, first projections) read nothing of one another -- one launch, the first `n_graph` workgroups build the
// lists, the others run the encoder's tiles on their first 64 * MFMA_WAVES threads (the other waves leave at once: a
// workgroup's barrier counts the waves that are still there).  Either body is the kernel's it comes from: the same bits.
DRP_GLOBAL void __launch_bounds__(GRAPH_Q4_THREADS)
km_graph_q4_encode(const float* __restrict__ s_prev, int prev_mod, size_t prev_stride, const float* __restrict__ s_delta, int N, int B,
                   int16_t* __restrict__ nbr_idx, uint8_t* __restrict__ nbr_cnt, DrpCam cam, float thr, int chunks, int self_first,
                   int n_graph, const uint16_t* __restrict__ sw6, const float* __restrict__ mw, const float* __restrict__ attr,
                   int attr_mod, const float* __restrict__ dens, int dens_mod, float* __restrict__ eff, float* __restrict__ c_node,
                   float* __restrict__ proj) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    static_assert(GRAPH_Q4_THREADS >= 64 * MFMA_WAVES, "the encoder's waves are the first of the workgroup");
    if ((int)blockIdx.x < n_graph) {
        graph_q4_block(s_prev, prev_mod, prev_stride, nullptr, 0, const_cast<float*>(s_delta), N, nbr_idx, nbr_cnt, cam, thr, chunks,
                       self_first, (int)blockIdx.x, lds);
        return;
    }
    if (threadIdx.x >= 64 * MFMA_WAVES) return;
    node_encode_split_block(sw6, mw, s_delta, attr, attr_mod, dens, dens_mod, N, B, eff, c_node, proj, (int)blockIdx.x - n_graph,
                            (int)gridDim.x - n_graph, lds);
}
#define KM_GRAPH_Q4_ENCODE_LDS(N) (GRAPH_Q4_LDS(N) > KM_NODE_SPLIT_LDS ? GRAPH_Q4_LDS(N) : KM_NODE_SPLIT_LDS)
