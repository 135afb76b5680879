// capi_train.h -- a section of the C ABI's translation unit (textually included by drp_capi.hip, in this order: capi_ctx.h,
// capi_pipeline.h, then inside extern "C": capi_core.h, capi_mpc.h, capi_prep.h, capi_gd.h, capi_train.h, capi_comm.h, capi_debug.h).
// Here: training on the same kernels (row f4): forward with tape, reverse mode with weight gradients, Adam, re-packing.

// ---- training on the same kernels (row f4) ------------------------------------------------------
namespace {
#define TR_GRAD_PAD ((W_TOTAL + 63) & ~63)      // tr_grad: the gradient blob, then (from here) kmb_step_bwd's barrier counters
// the batch as uploaded (drp_train_step): states [B][H+1][N][3] | impulses [B][H][N][3] | attributes [B][H+1][N] | densities [B]
// | particle counts [B] (ints), every block 16-byte aligned
struct TrArena { size_t states, sdelta, attrs, dens, nums, bytes; };
TrArena tr_layout(int B, int H, int N) {
    auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    TrArena a{};
    a.states = 0;
    a.sdelta = up(a.states + (size_t)B * (H + 1) * N * 3 * sizeof(float));
    a.attrs = up(a.sdelta + (size_t)B * H * N * 3 * sizeof(float));
    a.dens = up(a.attrs + (size_t)B * (H + 1) * N * sizeof(float));
    a.nums = up(a.dens + (size_t)B * sizeof(float));
    a.bytes = up(a.nums + (size_t)B * sizeof(int));
    return a;
}
const float* tr_given(const drp_ctx* c) { return static_cast<const float*>(c->tr_arena.p); }
const int* tr_nums(const drp_ctx* c, int B, int N) {
    return reinterpret_cast<const int*>(static_cast<const char*>(c->tr_arena.p) + tr_layout(B, c->tr_nroll, N).nums);
}
// forward over n_rollout steps (+ loss), optionally the backward pass with weight gradients
int train_forward_backward(drp_ctx* c, int B, int N, bool backward) {
    const int H = c->tr_nroll;
    const size_t bn = (size_t)B * N, bn64 = bn * 64, bnk = bn * DRP_K;
    const size_t hstride = (size_t)H * N * 3;                 // predicted states [B][H][N][3]
    const size_t in_stride = (size_t)(H + 1) * N * 3;         // given states     [B][H+1][N][3]
    hipStream_t st = c->stream;
    const bool rev_lds = N <= KB_REV_LDS_MAX_N && !c->rev_global_only;
    float* states = ptr<float>(c->states);
    const float* given = tr_given(c);
    const int* nums = tr_nums(c, B, N);
    float* eh = ptr<float>(c->eff_hist);
    unsigned* mh = ptr<unsigned>(c->tape_mask);
    float* ah = ptr<float>(c->agg_hist);
    float* g_state = ptr<float>(c->g_state);
    double* loss = c->tr_loss_host ? c->tr_loss_host : ptr<double>(c->tr_loss);      // pinned host memory: the terms land where the caller reads them
    const float scale = 1.0f / (float)(H * B);
    const int saved_engine = c->engine;
    c->engine = c->tr_engine;
    const float* cself = nullptr;
    const uint8_t* cself_ok = nullptr;
    int rc = prepare_cself(c, B, N, B, &cself, &cself_ok);
    for (int t = 0; t < H && rc == DRP_OK; ++t) {
        // with a backward pass to follow, the step's impulses and neighbour lists are part of the tape: its workspace
        // pointers are lent the tape's slices for the call (as the GD planner does) instead of three copies afterwards
        void* const save_sd = c->s_delta.p; void* const save_idx = c->nbr_idx.p; void* const save_cnt = c->nbr_cnt.p;
        struct Lend {
            drp_ctx* c; void* sd; void* idx; void* cnt;
            ~Lend() { c->s_delta.p = sd; c->nbr_idx.p = idx; c->nbr_cnt.p = cnt; }
        } lend{c, save_sd, save_idx, save_cnt};
        // this step's impulses are data (train/train_gnn_dyn.py:181): the step-major copy kt_unpack_inputs left (the tape's)
        c->s_delta.p = ptr<float>(c->tape_sdelta) + (size_t)t * bn * 3;
        if (backward) {
            c->nbr_idx.p = ptr<int16_t>(c->tape_idx) + (size_t)t * bnk;
            c->nbr_cnt.p = ptr<uint8_t>(c->tape_cnt) + (size_t)t * bn;
        }
        StepArgs a{};
        if (t == 0) { a.s_prev = given; a.prev_mod = B; a.prev_stride = in_stride; }
        else { a.s_prev = states + (size_t)(t - 1) * N * 3; a.prev_mod = B; a.prev_stride = hstride; }
        a.attr = ptr<float>(c->attr); a.attr_mod = B;
        a.dens = ptr<float>(c->dens); a.dens_mod = B;
        a.actions = nullptr; a.act_stride = 0;
        a.build_graph = true;
        a.s_out = states + (size_t)t * N * 3; a.out_stride = hstride;
        a.B = B; a.N = N;
        a.cself = cself; a.cself_ok = cself_ok;
        a.padded = true;                // collate_fn pads with zero rows: coincident particles
        if (backward) {
            a.eff_hist = eh + (size_t)t * 4 * bn64;
            a.mask_hist = mh + (size_t)t * DRP_PSTEP * bnk * 2;
            a.agg_hist = ah + (size_t)t * 3 * bn64;
        }
        rc = run_step(c, a);
        if (rc != DRP_OK) break;
    }
    c->engine = saved_engine;
    CHK(rc);
    // the loss of every step and d loss / d s_pred_t (train/train_gnn_dyn.py:184-186, :203) in one launch
    hipLaunchKernelGGL(kt_mse_grad, dim3(B, H), dim3(256), 0, st, states, hstride, given + (size_t)N * 3, in_stride,
                       nums, N, scale, g_state, loss, backward ? ptr<float>(c->tr_grad) : (float*)nullptr,
                       (size_t)TR_GRAD_PAD + (size_t)H * c->n_cu + 1);      // the gradient blob with kmb_step_bwd's counters behind it
    HIPCHK(c, hipGetLastError());
    if (!backward) return DRP_OK;

    const float* vw = ptr<float>(c->w_valu);
    const float* wraw = ptr<float>(c->w_raw);
    float* G = ptr<float>(c->tr_grad);
    // a training batch is a handful of samples: split each sample's rows over workgroups
    // (row kernels: one receiver per wave and pass; edge kernels: one receiver per 16 lanes)
    auto pick = [&](int rows_per_block) {
        int ch = (N + rows_per_block - 1) / rows_per_block;
        if (ch > 4096 / B) ch = 4096 / B;
        return ch < 1 ? 1 : ch;
    };
    const int chunks = pick(4), chunks16 = pick(16);
    const dim3 rgrid((unsigned)(B * chunks)), egrid((unsigned)(B * chunks16));
    const float* dens = ptr<float>(c->dens);
    // the reversed lists of ALL rollout steps in one launch (the tape holds every step's lists; a training batch is a handful
    // of workgroups per step)
    c->dv(N <= 512 ? DV_REV_256 : DV_REV_1024);
    if (N <= 512)
        hipLaunchKernelGGL(kb_reverse_lists<256>, dim3(B * H), dim3(256), KB_REV_LDS(N, rev_lds), st, ptr<int16_t>(c->tape_idx),
                           ptr<uint8_t>(c->tape_cnt), N, ptr<int>(c->rev_off), ptr<int>(c->rev), rev_lds ? 1 : 0, nums, B);
    else
        hipLaunchKernelGGL(kb_reverse_lists<1024>, dim3(B * H), dim3(1024), KB_REV_LDS(N, rev_lds), st, ptr<int16_t>(c->tape_idx),
                           ptr<uint8_t>(c->tape_cnt), N, ptr<int>(c->rev_off), ptr<int>(c->rev), rev_lds ? 1 : 0, nums, B);
    // deferred weight gradients: what a job reads keeps a buffer per rollout step t (g_eff and g_proj: per propagation step
    // too; slot 0 of g_eff is the transient copy the predictor writes and the particle encoder reads)
    const bool defer = c->wg_defer_now;
    const size_t per_t = defer ? 1 : 0;
    // The node stages of a rollout step in ONE launch (kmb_step_bwd<dump>): it needs every dump in a buffer of its own (the
    // deferred weight gradients' layout).  A group of `f_spw` samples is shared by `f_parts` workgroups, the grid at most one
    // workgroup per CU (the kernel's barrier in memory).  It pays for a handful of tiles only (the reference's batch of
    // 4 x <= 300 particles: 40): every tile is a chain of memory round trips, and the stage kernels spread the same gathers
    // over more threads (32 x 300: 2.6 ms per iteration staged, 4.3 in one launch)
    const long f_tiles = (long)B * ((N + 31) / 32);
    const bool fused = defer && c->bwd_fused && !c->bwd_valu_stages &&
                       (c->train_fused >= 0 ? c->train_fused != 0 : f_tiles <= c->n_cu / 4);
    const int f_spw = (B + c->n_cu - 1) / c->n_cu, f_groups = (B + f_spw - 1) / f_spw;
    int f_parts = 1;
    bool f_coop = false;
    if (fused) {
        const int group_tiles = (int)(((long)f_spw * N + 31) / 32);
        f_parts = c->train_parts > 0 ? c->train_parts : c->n_cu / f_groups;
        if (f_parts > c->n_cu / f_groups) f_parts = c->n_cu / f_groups;
        if (f_parts > group_tiles) f_parts = group_tiles;
        if (f_parts < 1) f_parts = 1;
        // a handful of tiles per workgroup: all eight waves gather a tile's edge terms (a wave on its own is one long chain of
        // L2 round trips per tile and phase: 27 us against 6); many: a tile per wave, the waves hide each other's latency
        f_coop = c->train_coop >= 0 ? c->train_coop != 0 : (group_tiles + f_parts - 1) / f_parts <= 2 * KMB_COOP_SLOTS;
    }
    unsigned* const f_bar = reinterpret_cast<unsigned*>(G + TR_GRAD_PAD);        // [H][f_groups] arrival counters, then the give-up flag
    for (int t = H - 1; t >= 0; --t) {
        const size_t tt = per_t * (size_t)t;
        float* const ge_tmp = ptr<float>(c->g_eff);
        auto ge_v = [&](int v) { return ptr<float>(c->g_eff) + per_t * ((size_t)(t * 3 + v) + 1) * bn64; };   // v = 0, 1, 2: steps 2, 1, 0
        auto gp_v = [&](int p) { return ptr<float>(c->g_proj) + per_t * (size_t)(t * 3 + p) * bn64 * 2; };
        float* const g_cnode_t = ptr<float>(c->g_cnode) + tt * bn64;
        float* const tr_hact_t = ptr<float>(c->tr_hact) + tt * bn64;
        float* const tr_gh_t = ptr<float>(c->tr_gh) + tt * bn64;
        float* const tr_gpe_t = ptr<float>(c->tr_gpe) + tt * bn64;
        float* const tr_a1n_t = ptr<float>(c->tr_a1n) + tt * bn64;
        float* const tr_gh1_t = ptr<float>(c->tr_gh1) + tt * bn64;
        float* const tr_xn_t = ptr<float>(c->tr_xn) + tt * bn * 8;
        KbEdgeDump ed{ptr<float>(c->ed_re) + tt * bnk * 64, ptr<float>(c->ed_a2) + tt * bnk * 64, ptr<float>(c->ed_a1) + tt * bnk * 64,
                      ptr<float>(c->ed_x0) + tt * bnk * 8, ptr<float>(c->ed_gce) + tt * bnk * 64, ptr<float>(c->ed_g3) + tt * bnk * 64,
                      ptr<float>(c->ed_g2) + tt * bnk * 64, ptr<float>(c->ed_g1) + tt * bnk * 64};
        const float* s_prev = (t == 0) ? given : states + (size_t)(t - 1) * N * 3;
        const size_t prev_stride = (t == 0) ? in_stride : hstride;
        float* eht = eh + (size_t)t * 4 * bn64;
        const unsigned* mht = mh + (size_t)t * DRP_PSTEP * bnk * 2;
        float* aht = ah + (size_t)t * 3 * bn64;
        const int16_t* idx = ptr<int16_t>(c->tape_idx) + (size_t)t * bnk;
        const uint8_t* cnt = ptr<uint8_t>(c->tape_cnt) + (size_t)t * bn;
        float* g_out = g_state + (size_t)t * bn * 3;
        float* gah = ptr<float>(c->g_agg_hist);
        int* const rev_off_t = ptr<int>(c->rev_off) + (size_t)t * B * (N + 1);
        int* const rev_t = ptr<int>(c->rev) + (size_t)t * bnk;
        // node-level stages: on the matrix cores when the batch has enough 32-row tiles to fill the chip,
        // otherwise the row kernels chunked over (sample, rows)
        if (fused) {
            // everything between the loss gradient and the relation encoder's backward in ONE launch (kmb_step_bwd<DUMP>): the
            // operands of the weight gradients are its dumps; same queue order as the stage kernels below
            c->dv(f_coop ? DV_TRAIN_NODE_FUSED_COOP : DV_TRAIN_NODE_FUSED);
            KmbDump dump{};
            dump.hact = tr_hact_t; dump.gh = tr_gh_t;
            for (int v = 0; v < 3; ++v) { dump.ge[v] = ge_v(v); dump.gp[v] = gp_v(v); }
            dump.gpe = tr_gpe_t; dump.a1n = tr_a1n_t; dump.gh1 = tr_gh1_t; dump.xn = tr_xn_t;
#define STEP_BWD_ARGS ptr<float>(c->w_mfma), ptr<float>(c->w_mfma_bwd), eht, mht, cnt, rev_off_t, rev_t, g_out, (size_t)N * 3, \
                      ptr<float>(c->tape_sdelta) + (size_t)t * bn * 3, ptr<float>(c->attr), B, dens, B, N, B, f_spw, ge_tmp, \
                      g_cnode_t, gah, ptr<float>(c->g_sdelta), dump, f_parts, f_bar + (size_t)t * f_groups, f_bar + (size_t)H * f_groups
            if (f_coop)
                hipLaunchKernelGGL((kmb_step_bwd<true, true>), dim3((unsigned)(f_groups * f_parts)), dim3(64 * KMB_FUSED_WAVES), KMB_COOP_LDS, st, STEP_BWD_ARGS);
            else
                hipLaunchKernelGGL((kmb_step_bwd<true, false>), dim3((unsigned)(f_groups * f_parts)), dim3(64 * KMB_FUSED_WAVES), KMB_FUSED_LDS, st, STEP_BWD_ARGS);
#undef STEP_BWD_ARGS
            launch_wgrad<64>(c, tr_gh_t, 64, eht + 3 * bn64, 64, (long)bn, G + W_PR0_W, 64, 1, G + W_PR0_B, nullptr, nullptr, 1, 1);
            launch_wgrad<3>(c, tr_hact_t, 64, g_out, 3, (long)bn, G + W_PR1_W, 1, 64, nullptr, nullptr, nullptr, 1, 1);
            for (int p = DRP_PSTEP - 1; p >= 0; --p) {
                launch_wgrad<64>(c, ge_v(DRP_PSTEP - 1 - p), 64, aht + (size_t)p * bn64, 64, (long)bn, G + W_PP_W + 64, 129, 1,
                                 nullptr, nullptr, nullptr, 1, 1);
                launch_wgrad<64>(c, gp_v(p), 128, eht + (size_t)p * bn64, 64, (long)bn, G + W_RP_W + 64, 193, 1,
                                 nullptr, nullptr, nullptr, 1, 1);
                launch_wgrad<64>(c, gp_v(p) + 64, 128, eht + (size_t)p * bn64, 64, (long)bn, G + W_RP_W + 128, 193,
                                 1, nullptr, nullptr, nullptr, 1, 1);
            }
            launch_wgrad<64>(c, g_cnode_t, 64, eht, 64, (long)bn, G + W_PP_W, 129, 1, G + W_PP_B, G + W_PP_W + 128,
                             dens, B, (long)N);
            launch_wgrad<64>(c, tr_gpe_t, 64, tr_a1n_t, 64, (long)bn, G + W_PE2_W, 64, 1, G + W_PE2_B,
                             nullptr, nullptr, 1, 1);
            launch_wgrad<5>(c, tr_gh1_t, 64, tr_xn_t, 8, (long)bn, G + W_PE0_W, 5, 1, G + W_PE0_B,
                            nullptr, nullptr, 1, 1);
        } else if ((long)B * ((N + 31) / 32) >= KMB_MIN_TILES && !c->bwd_valu_stages) {
            const float* mw = ptr<float>(c->w_mfma);
            const float* mb = ptr<float>(c->w_mfma_bwd);
            const long node_tiles = (long)B * ((N + 31) / 32);
            const dim3 ngrid(mfma_grid_spread(c, node_tiles)), nblk(64 * MFMA_WAVES);
            c->dv(DV_TRAIN_NODE_MFMA);
            // predictor
            hipLaunchKernelGGL(kmb_predict, ngrid, nblk, KMB_PREDICT_LDS, st, mw, mb, eht + 3 * bn64, g_out, (size_t)N * 3, N, B,
                               ge_tmp, tr_hact_t, tr_gh_t);
            launch_wgrad<64>(c, tr_gh_t, 64, eht + 3 * bn64, 64, (long)bn, G + W_PR0_W, 64, 1, G + W_PR0_B, nullptr,
                             nullptr, 1, 1);
            launch_wgrad<3>(c, tr_hact_t, 64, g_out, 3, (long)bn, G + W_PR1_W, 1, 64, nullptr, nullptr, nullptr, 1, 1);
            // update of the last propagation step; then per step the edge terms and, in one launch, the
            // projection of this step with the update of the one before (k_backward_mfma.h)
            hipLaunchKernelGGL((kmb_node_step<false, true>), ngrid, nblk, KMB_STEP_LDS(false, true), st, mb, ge_tmp, ge_v(0),
                               (const float*)nullptr, eht + (size_t)DRP_PSTEP * bn64, g_cnode_t, 1,
                               gah + (size_t)(DRP_PSTEP - 1) * bn64, N, B);
            for (int p = DRP_PSTEP - 1; p >= 0; --p) {
                float* g_agg_p = gah + (size_t)p * bn64;
                const unsigned* mask_p = mht + (size_t)p * bnk * 2;
                float* const ge_p = ge_v(DRP_PSTEP - 1 - p);     // the pre-activation gradient of step p
                float* const gp_p = gp_v(p);
                // particle propagator, aggregate columns
                launch_wgrad<64>(c, ge_p, 64, aht + (size_t)p * bn64, 64, (long)bn, G + W_PP_W + 64, 129, 1,
                                 nullptr, nullptr, nullptr, 1, 1);
                hipLaunchKernelGGL(kb_edge_terms, egrid, dim3(256), 0, st, g_agg_p, mask_p, cnt, rev_off_t,
                                   rev_t, N, gp_p, chunks16);
                // relation propagator, receiver and sender columns
                launch_wgrad<64>(c, gp_p, 128, eht + (size_t)p * bn64, 64, (long)bn, G + W_RP_W + 64, 193, 1,
                                 nullptr, nullptr, nullptr, 1, 1);
                launch_wgrad<64>(c, gp_p + 64, 128, eht + (size_t)p * bn64, 64, (long)bn, G + W_RP_W + 128, 193,
                                 1, nullptr, nullptr, nullptr, 1, 1);
                flush_wgrad(c);                          // (not deferred:) before the next kernel overwrites g_eff (and, next step, g_proj)
                if (p > 0)
                    hipLaunchKernelGGL((kmb_node_step<true, true>), ngrid, nblk, KMB_STEP_LDS(true, true), st, mb,
                                       ge_p, ge_v(DRP_PSTEP - p), gp_p, eht + (size_t)p * bn64,
                                       g_cnode_t, 0, gah + (size_t)(p - 1) * bn64, N, B);
                else
                    hipLaunchKernelGGL((kmb_node_step<true, false>), ngrid, nblk, KMB_STEP_LDS(true, false), st, mb,
                                       ge_p, ge_tmp, gp_p, (const float*)nullptr, (float*)nullptr, 0,
                                       (float*)nullptr, N, B);
            }
            // particle propagator, encoder columns + density column + bias; particle encoder
            launch_wgrad<64>(c, g_cnode_t, 64, eht, 64, (long)bn, G + W_PP_W, 129, 1, G + W_PP_B, G + W_PP_W + 128,
                             dens, B, (long)N);
            hipLaunchKernelGGL(kmb_node_encode, ngrid, nblk, KMB_NODE_ENCODE_LDS, st, mw, mb,
                               ptr<float>(c->tape_sdelta) + (size_t)t * bn * 3, ptr<float>(c->attr), B, dens, B, eht,
                               ge_tmp, g_cnode_t, N, B, ptr<float>(c->g_sdelta), tr_gpe_t,
                               tr_a1n_t, tr_gh1_t, tr_xn_t);
            launch_wgrad<64>(c, tr_gpe_t, 64, tr_a1n_t, 64, (long)bn, G + W_PE2_W, 64, 1, G + W_PE2_B,
                             nullptr, nullptr, 1, 1);
            launch_wgrad<5>(c, tr_gh1_t, 64, tr_xn_t, 8, (long)bn, G + W_PE0_W, 5, 1, G + W_PE0_B,
                            nullptr, nullptr, 1, 1);
            flush_wgrad(c);
        } else {
            c->dv(DV_TRAIN_NODE_VALU);
            // predictor
            hipLaunchKernelGGL(kb_predict, rgrid, dim3(256), 0, st, vw, wraw, eht + 3 * bn64, g_out, (size_t)N * 3, N,
                               ptr<float>(c->g_eff), ptr<float>(c->tr_hact), ptr<float>(c->tr_gh), chunks);
            launch_wgrad<64>(c, ptr<float>(c->tr_gh), 64, eht + 3 * bn64, 64, (long)bn, G + W_PR0_W, 64, 1, G + W_PR0_B, nullptr,
                             nullptr, 1, 1);
            launch_wgrad<3>(c, ptr<float>(c->tr_hact), 64, g_out, 3, (long)bn, G + W_PR1_W, 1, 64, nullptr, nullptr, nullptr, 1, 1);
            for (int p = DRP_PSTEP - 1; p >= 0; --p) {
                float* g_agg_p = gah + (size_t)p * bn64;
                const unsigned* mask_p = mht + (size_t)p * bnk * 2;
                hipLaunchKernelGGL(kb_update, rgrid, dim3(256), 0, st, wraw, eht + (size_t)(p + 1) * bn64,
                                   ptr<float>(c->g_eff), ptr<float>(c->g_cnode), p == DRP_PSTEP - 1 ? 1 : 0, N, g_agg_p, chunks);
                // particle propagator, aggregate columns: g_eff now holds the pre-activation gradient
                launch_wgrad<64>(c, ptr<float>(c->g_eff), 64, aht + (size_t)p * bn64, 64, (long)bn, G + W_PP_W + 64, 129, 1,
                                 nullptr, nullptr, nullptr, 1, 1);
                hipLaunchKernelGGL(kb_edge_terms, egrid, dim3(256), 0, st, g_agg_p, mask_p, cnt, rev_off_t,
                                   rev_t, N, ptr<float>(c->g_proj), chunks16);
                // relation propagator, receiver and sender columns
                launch_wgrad<64>(c, ptr<float>(c->g_proj), 128, eht + (size_t)p * bn64, 64, (long)bn, G + W_RP_W + 64, 193, 1,
                                 nullptr, nullptr, nullptr, 1, 1);
                launch_wgrad<64>(c, ptr<float>(c->g_proj) + 64, 128, eht + (size_t)p * bn64, 64, (long)bn, G + W_RP_W + 128, 193,
                                 1, nullptr, nullptr, nullptr, 1, 1);
                flush_wgrad(c);                          // before kb_project overwrites g_eff
                hipLaunchKernelGGL(kb_project, rgrid, dim3(256), 0, st, wraw, ptr<float>(c->g_proj), N, ptr<float>(c->g_eff), chunks);
            }
            // particle propagator, encoder columns + density column + bias; particle encoder
            launch_wgrad<64>(c, ptr<float>(c->g_cnode), 64, eht, 64, (long)bn, G + W_PP_W, 129, 1, G + W_PP_B, G + W_PP_W + 128,
                             dens, B, (long)N);
            hipLaunchKernelGGL(kb_node_encode, rgrid, dim3(256), 0, st, vw, wraw,
                               ptr<float>(c->tape_sdelta) + (size_t)t * bn * 3, ptr<float>(c->attr), B, dens, B, eht,
                               ptr<float>(c->g_eff), ptr<float>(c->g_cnode), N, ptr<float>(c->g_sdelta), ptr<float>(c->tr_gpe),
                               ptr<float>(c->tr_a1n), ptr<float>(c->tr_gh1), ptr<float>(c->tr_xn), chunks);
            launch_wgrad<64>(c, ptr<float>(c->tr_gpe), 64, ptr<float>(c->tr_a1n), 64, (long)bn, G + W_PE2_W, 64, 1, G + W_PE2_B,
                             nullptr, nullptr, 1, 1);
            launch_wgrad<5>(c, ptr<float>(c->tr_gh1), 64, ptr<float>(c->tr_xn), 8, (long)bn, G + W_PE0_W, 5, 1, G + W_PE0_B,
                            nullptr, nullptr, 1, 1);
            flush_wgrad(c);
        }
        // the previous step's output feeds this step as s_cur: residual + relation encoder
        float* g_prev = nullptr;
        if (t > 0) {
            g_prev = g_state + (size_t)(t - 1) * bn * 3;
            // the residual's share: with the matrix-core edge kernel nothing touches g_prev before kb_gather_pos, which adds it first
            if (!c->bwd_edge_mfma)
                hipLaunchKernelGGL(kt_add, dim3((unsigned)((bn * 3 + 255) / 256)), dim3(256), 0, st, g_prev, g_out, bn * 3);
        }
        c->dv(c->bwd_edge_mfma ? DV_BWD_EDGE_MFMA : DV_BWD_EDGE_VALU);
        if (c->bwd_edge_mfma)
            launch_edge_encode_mfma(c, s_prev, B, prev_stride, B, idx, cnt, gah, mht, bn, N, B,
                                    g_prev != nullptr ? ptr<float>(c->gpos_edge) : (float*)nullptr, ed);
        else
            hipLaunchKernelGGL(kb_edge_encode, rgrid, dim3(256), KB_EDGE_ENCODE_LDS, st, vw, wraw, s_prev, B,
                               prev_stride, ptr<float>(c->attr), B, dens, B, idx, cnt, gah, mht, bn, N, g_prev, (size_t)N * 3, ptr<float>(c->gpos_edge), ed, chunks);
        if (g_prev != nullptr)
            hipLaunchKernelGGL(kb_gather_pos, dim3((N + 255) / 256, B), dim3(256), 0, st, ptr<float>(c->gpos_edge),
                               rev_off_t, rev_t, N, g_prev, (size_t)N * 3, c->bwd_edge_mfma ? 1 : 0, cnt,
                               c->bwd_edge_mfma ? (const float*)g_out : (const float*)nullptr);
        launch_wgrad<64>(c, ed.gce, 64, ed.re, 64, (long)bnk, G + W_RP_W, 193, 1, G + W_RP_B, G + W_RP_W + 192, dens, B,
                         (long)N * DRP_K);
        launch_wgrad<64>(c, ed.g3, 64, ed.a2, 64, (long)bnk, G + W_RE4_W, 64, 1, G + W_RE4_B, nullptr, nullptr, 1, 1);
        launch_wgrad<64>(c, ed.g2, 64, ed.a1, 64, (long)bnk, G + W_RE2_W, 64, 1, G + W_RE2_B, nullptr, nullptr, 1, 1);
        launch_wgrad<6>(c, ed.g1, 64, ed.x0, 8, (long)bnk, G + W_RE0_W, 6, 1, G + W_RE0_B, nullptr, nullptr, 1, 1);
        flush_wgrad(c);                                  // the next rollout step rewrites the dumps these jobs read
    }
    flush_wgrad(c);
    // bias of the predictor's last layer: the column sums of every step's d loss / d s_pred (all final by now), one launch
    hipLaunchKernelGGL(kt_colsum3, dim3(1), dim3(1024), 0, st, g_state, (long)(H * bn), G + W_PR1_B);
    if (defer) CHK(flush_wgrad_all(c));
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

// The packed copies of the weights follow an optimiser step without a round trip of the packers through the host
// (k_train.h): only the blob itself comes back -- the split relation encoder's range shift is a function of the
// weights (set_split_range), and drp_get_weights serves the host copy.
int ensure_repack_maps(drp_ctx* c) {
    if (c->repack_maps_ready) return DRP_OK;
    std::vector<float> probe((size_t)W_TOTAL);
    for (int i = 0; i < (int)W_TOTAL; ++i) probe[i] = (float)(i + 1);          // exact in fp32 (38 403 < 2^24)
    auto to_map = [](const std::vector<float>& packed) {
        std::vector<int> m(packed.size());
        for (size_t i = 0; i < packed.size(); ++i)
            m[i] = packed[i] == 0.0f ? 0 : (packed[i] < 0.0f ? -1 : (int)packed[i]);
        return m;
    };
    std::vector<float> v, m, mb;
    pack_valu(probe.data(), v);
    pack_mfma(probe.data(), m);
    pack_mfma_bwd(probe.data(), mb);
    const std::vector<int> mv = to_map(v), mm = to_map(m), mmb = to_map(mb);
    CHK(h2d(c, c->map_valu, mv.data(), mv.size() * sizeof(int)));
    CHK(h2d(c, c->map_mfma, mm.data(), mm.size() * sizeof(int)));
    CHK(h2d(c, c->map_mfma_bwd, mmb.data(), mmb.size() * sizeof(int)));
    CHK(guarded_wait(c, nullptr));                   // the vectors go out of scope
    if (!c->w_pin) HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->w_pin), ((size_t)W_TOTAL + 4) * sizeof(float), hipHostMallocDefault));
    c->repack_maps_ready = true;
    return DRP_OK;
}

// launches only: the caller's final wait brings the blob and the device's shift back (finish_repack)
int repack_on_device(drp_ctx* c, bool blob_in_pin = false /* k_adam has written the updated blob to w_pin already */) {
    CHK(ensure_repack_maps(c));
    CHK(ensure(c, c->re_shift_dev, sizeof(int)));
    hipStream_t st = c->stream;
    const float* w = ptr<float>(c->w_raw);
    RepackAll a{};
    a.map_v = ptr<int>(c->map_valu); a.dst_v = ptr<float>(c->w_valu); a.n_v = (int)V_TOTAL;
    a.map_m = ptr<int>(c->map_mfma); a.dst_m = ptr<float>(c->w_mfma); a.n_m = (int)M_TOTAL;
    a.map_mb = ptr<int>(c->map_mfma_bwd); a.dst_mb = ptr<float>(c->w_mfma_bwd); a.n_mb = (int)MB_TOTAL;
    a.out6 = ptr<uint16_t>(c->w_split6); a.out6b = ptr<uint16_t>(c->w_split6_bwd);
    a.forced_shift = c->re_shift_env; a.shift_out = ptr<int>(c->re_shift_dev);
    hipLaunchKernelGGL(kt_repack_all, dim3(KT_REPACK_ALL_BLOCKS((int)V_TOTAL, (int)M_TOTAL, (int)MB_TOTAL)), dim3(256), 0, st, w, a);
    // the relation encoder's range shift depends on the new weights: the launch above derived it; the blob itself comes
    // back too (it is the host copy drp_get_weights serves, and the host's own range for the calls to come)
    hipLaunchKernelGGL(kt_repack_split, dim3(4 * 16), dim3(256), 0, st, w, 0, ptr<uint16_t>(c->w_split), ptr<int>(c->re_shift_dev),
                       blob_in_pin ? reinterpret_cast<int*>(c->w_pin + W_TOTAL) : (int*)nullptr);
    if (!blob_in_pin) {
        HIPCHK(c, hipMemcpyAsync(c->w_pin, c->w_raw.p, (size_t)W_TOTAL * sizeof(float), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(c->w_pin + W_TOTAL, c->re_shift_dev.p, sizeof(int), hipMemcpyDeviceToHost, st));
    }
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}
// after the wait: the host's copy of the blob and its range; host and device derive the shift by the same operations --
// should they ever disagree, the fragments are packed again with the host's
int finish_repack(drp_ctx* c) {
    c->w_host.assign(c->w_pin, c->w_pin + W_TOTAL);
    set_split_range(c, c->w_host.data());
    int dev_shift;
    memcpy(&dev_shift, c->w_pin + W_TOTAL, sizeof(int));
    if (dev_shift != c->re_range.shift) {
        hipLaunchKernelGGL(kt_repack_split, dim3(4 * 16), dim3(256), 0, c->stream, ptr<float>(c->w_raw), c->re_range.shift,
                           ptr<uint16_t>(c->w_split), (const int*)nullptr);
        HIPCHK(c, hipGetLastError());
        CHK(guarded_wait(c, nullptr));
    }
    return DRP_OK;
}

int install_weights(drp_ctx* c, const std::vector<float>& blob) {
    std::vector<float> tmp(blob);
    return drp_load_weights(c, tmp.data(), tmp.size(), c->adj_thresh);
}
}  // namespace

int drp_train_begin(drp_ctx* c, int n_rollout, double lr, double beta1) {
    CHK(need(c, true, false, false));
    if (n_rollout < 1 || n_rollout > 64 || !(lr > 0.0) || !(beta1 >= 0.0 && beta1 < 1.0))
        return fail(c, DRP_EINVAL, "bad training arguments n_rollout=%d lr=%g beta1=%g", n_rollout, lr, beta1);
    HIPCHK(c, hipSetDevice(c->device));
    CHK(ensure(c, c->tr_grad, ((size_t)TR_GRAD_PAD + (size_t)n_rollout * c->n_cu + 1) * sizeof(float)));
    CHK(ensure(c, c->tr_m, (size_t)W_TOTAL * sizeof(float)));
    CHK(ensure(c, c->tr_v, (size_t)W_TOTAL * sizeof(float)));
    CHK(ensure(c, c->tr_part, (size_t)KT_WGRAD_MAX_JOBS * KT_WGRAD_MAX_BLOCKS * 66 * 64 * sizeof(float)));
    HIPCHK(c, hipMemsetAsync(c->tr_m.p, 0, (size_t)W_TOTAL * sizeof(float), c->stream));
    HIPCHK(c, hipMemsetAsync(c->tr_v.p, 0, (size_t)W_TOTAL * sizeof(float), c->stream));
    CHK(guarded_wait(c, nullptr));
    c->tr_nroll = n_rollout; c->tr_lr = lr; c->tr_beta1 = beta1; c->tr_iter = 0;
    c->tr_on = true;
    c->gd_on = false;
    c->mpc_on = false;
    return DRP_OK;
}

int drp_train_step(drp_ctx* c, const float* states, const float* states_delta, const float* attrs,
                   const int32_t* particle_nums, const float* particle_dens, int B, int N, int mode, double* loss_out,
                   float* grad_out) {
    if (!c || !c->tr_on) return fail(c, DRP_ESTATE, "drp_train_begin not called");
    CHK(check_bn(c, B, N));
    if (!states || !states_delta || !attrs || !particle_nums || !particle_dens) return fail(c, DRP_EINVAL, "null argument");
    if (mode < DRP_TRAIN_EVAL || mode > DRP_TRAIN_UPDATE) return fail(c, DRP_EINVAL, "bad mode %d", mode);
    for (int b = 0; b < B; ++b)
        if (particle_nums[b] <= 0 || particle_nums[b] > N)
            return fail(c, DRP_EINVAL, "particle_nums[%d]=%d outside 1..%d", b, particle_nums[b], N);
    HIPCHK(c, hipSetDevice(c->device));
    end_sessions(c);
    {
        float amax = 0.0f;                                    // a_cur = attrs[:, 0]
        for (int b = 0; b < B; ++b) amax = fmaxf(amax, max_abs(attrs + (size_t)b * (c->tr_nroll + 1) * N, (size_t)N));
        CHK(pick_tape_engine(c, amax, max_abs(particle_dens, (size_t)B), max_abs(states_delta, (size_t)B * c->tr_nroll * N * 3), &c->tr_engine));
    }
    const int H = c->tr_nroll;
    const size_t bn = (size_t)B * N, bn64 = bn * 64, bnk = bn * DRP_K;
    const bool backward = mode != DRP_TRAIN_EVAL;
    // the batch in one copy: packed into pinned staging in the caller's layouts, unpacked by one launch (kt_unpack_inputs)
    const TrArena lay = tr_layout(B, H, N);
    // behind the batch: what comes BACK after the one wait -- the loss terms [H][B] and the give-up flag of kmb_step_bwd's
    // barrier (pinned: the copies are asynchronous, nothing on the way touches pageable memory or this frame)
    const size_t back_off = lay.bytes, back_bytes = (size_t)H * B * sizeof(double) + 16;
    if (c->tr_pin_cap < lay.bytes + back_bytes) {
        if (c->tr_pin) {
            // kernels read and write this buffer directly: nothing of an earlier call (one that returned on an error before its
            // wait, say) may still be in flight when it goes
            (void)hipStreamSynchronize(c->stream);
            (void)hipHostFree(c->tr_pin); c->tr_pin = nullptr; c->tr_pin_cap = 0;
        }
        HIPCHK(c, hipHostMalloc(&c->tr_pin, lay.bytes + back_bytes, hipHostMallocDefault));
        c->tr_pin_cap = lay.bytes + back_bytes;
    }
    {
        char* pin = static_cast<char*>(c->tr_pin);
        memcpy(pin + lay.states, states, (size_t)B * (H + 1) * N * 3 * sizeof(float));
        memcpy(pin + lay.sdelta, states_delta, (size_t)B * H * N * 3 * sizeof(float));
        memcpy(pin + lay.attrs, attrs, (size_t)B * (H + 1) * N * sizeof(float));
        memcpy(pin + lay.dens, particle_dens, (size_t)B * sizeof(float));
        memcpy(pin + lay.nums, particle_nums, (size_t)B * sizeof(int));
    }
    CHK(ensure(c, c->tr_arena, lay.bytes));
    CHK(ensure(c, c->attr, bn * sizeof(float)));
    CHK(ensure(c, c->dens, (size_t)B * sizeof(float)));
    CHK(ensure(c, c->tape_sdelta, (size_t)H * bn * 3 * sizeof(float)));
    {
        // the unpacking launch IS the upload: it reads the staged batch from the pinned host buffer (device-visible) and leaves
        // the arena copy for the kernels that read the given states; DRP_TRAIN_COPY_UPLOAD=1: a copy on the stream first
        const bool by_kernel = !c->train_copy_upload;
        if (!by_kernel) CHK(h2d(c, c->tr_arena, c->tr_pin, lay.bytes));
        const char* ar = by_kernel ? static_cast<const char*>(c->tr_pin) : static_cast<const char*>(c->tr_arena.p);
        const size_t total = (size_t)H * bn * 3;
        hipLaunchKernelGGL(kt_unpack_inputs, dim3((unsigned)std::min<size_t>((total + 255) / 256, 1024)), dim3(256), 0, c->stream,
                           reinterpret_cast<const float*>(ar + lay.sdelta), reinterpret_cast<const float*>(ar + lay.attrs),
                           reinterpret_cast<const float*>(ar + lay.dens), B, H, N, ptr<float>(c->tape_sdelta), ptr<float>(c->attr),
                           ptr<float>(c->dens), by_kernel ? reinterpret_cast<const float4*>(c->tr_pin) : (const float4*)nullptr,
                           by_kernel ? static_cast<float4*>(c->tr_arena.p) : (float4*)nullptr, by_kernel ? lay.bytes / 16 : (size_t)0);
    }
    CHK(ensure_step_ws(c, B, N, c->tr_engine));
    CHK(ensure(c, c->states, (size_t)H * bn * 3 * sizeof(float)));
    CHK(ensure(c, c->g_state, (size_t)H * bn * 3 * sizeof(float)));
    c->wg_defer_now = false;
    c->wg_jobs.clear();
    bool defer_batch = false;               // re-armed before every pass of the step (flush_wgrad_all clears wg_defer_now)
    if (backward) {
        // deferred weight gradients keep every job's operands until the end of the backward pass: H copies of the node-level
        // dumps (3 H + 1 of g_eff, 3 H of g_proj) and of the relation encoder's dumps -- 0.24 GB per rollout step at 32 x 300
        const size_t keep_bytes = (size_t)H * (16 * bn64 + 7 * bnk * 64 + bnk * 8 + bn * 8) * sizeof(float);
        const bool defer = c->wgrad_defer && (long)B * ((N + 31) / 32) >= KMB_MIN_TILES && !c->bwd_valu_stages && keep_bytes <= ((size_t)8 << 30);
        c->wg_defer_now = defer;
        defer_batch = defer;
        const size_t kt = defer ? (size_t)H : 1;
        CHK(ensure(c, c->eff_hist, (size_t)H * 4 * bn64 * sizeof(float)));
        CHK(ensure(c, c->agg_hist, (size_t)H * 3 * bn64 * sizeof(float)));
        CHK(ensure(c, c->tape_sdelta, (size_t)H * bn * 3 * sizeof(float)));
        CHK(ensure(c, c->tape_idx, (size_t)H * bnk * sizeof(int16_t)));
        CHK(ensure(c, c->tape_cnt, (size_t)H * bn));
        CHK(ensure(c, c->tape_mask, (size_t)H * DRP_PSTEP * bnk * 2 * sizeof(unsigned)));
        CHK(ensure(c, c->g_agg_hist, (size_t)DRP_PSTEP * bn64 * sizeof(float)));
        CHK(ensure(c, c->rev_off, (size_t)H * B * (N + 1) * sizeof(int)));
        CHK(ensure(c, c->rev, (size_t)H * bnk * sizeof(int)));
        CHK(ensure(c, c->gpos_edge, bnk * 4 * sizeof(float)));
        CHK(ensure(c, c->g_eff, (defer ? 3 * kt + 1 : 1) * bn64 * sizeof(float)));
        CHK(ensure(c, c->g_cnode, kt * bn64 * sizeof(float)));
        CHK(ensure(c, c->g_agg, bn64 * sizeof(float)));
        CHK(ensure(c, c->g_proj, (defer ? 3 * kt : 1) * bn64 * 2 * sizeof(float)));
        CHK(ensure(c, c->g_sdelta, bn * 3 * sizeof(float)));
        DevBuf* node64[] = {&c->tr_hact, &c->tr_gh, &c->tr_gpe, &c->tr_a1n, &c->tr_gh1};
        for (DevBuf* b : node64) CHK(ensure(c, *b, kt * bn64 * sizeof(float)));
        CHK(ensure(c, c->tr_xn, kt * bn * 8 * sizeof(float)));
        DevBuf* edge64[] = {&c->ed_re, &c->ed_a2, &c->ed_a1, &c->ed_gce, &c->ed_g3, &c->ed_g2, &c->ed_g1};
        for (DevBuf* b : edge64) CHK(ensure(c, *b, kt * bnk * 64 * sizeof(float)));
        CHK(ensure(c, c->ed_x0, kt * bnk * 8 * sizeof(float)));
    }
    CHK(ensure(c, c->tr_loss, (size_t)H * B * sizeof(double)));
    c->lastH = H;
    double* const parts = reinterpret_cast<double*>(static_cast<char*>(c->tr_pin) + back_off);
    unsigned* const gave_up = reinterpret_cast<unsigned*>(parts + (size_t)H * B);
    // kmb_step_bwd's barrier among the workgroups of a group gives up after two seconds (k_backward_mfma.h) and sets a flag
    // behind its counters; the gradient of such a pass is partial.  The optimiser step reads the flag ON THE DEVICE and moves
    // nothing when it is set (k_adam's `skip`), the iteration count advances only once the flag has come back clear, and the
    // step runs again with one workgroup per group (no barrier to wait at) -- for the rest of the context's life.
    c->tr_loss_host = (loss_out && !c->train_copy_upload) ? parts : nullptr;
    struct LossHostReset { drp_ctx* c; ~LossHostReset() { c->tr_loss_host = nullptr; } } loss_host_reset{c};
    for (int attempt = 0; ; ++attempt) {
        c->wg_defer_now = defer_batch;
        c->wg_jobs.clear();
        CHK(train_forward_backward(c, B, N, backward));
        const int f_spw = (B + c->n_cu - 1) / c->n_cu, f_groups = (B + f_spw - 1) / f_spw;
        const unsigned* const flag_dev = reinterpret_cast<const unsigned*>(ptr<float>(c->tr_grad) + TR_GRAD_PAD + (size_t)H * f_groups);
        *gave_up = 0;
        // update iterations end WITHOUT a copy on the stream: the loss terms are stored to pinned host memory by the loss kernel,
        // the optimiser step writes the updated blob and the barrier's flag there, kt_repack_split the range shift -- a copy
        // engine's transfer between kernels costs tens of microseconds of hand-over (tools/train_trace.sh)
        const bool direct = mode == DRP_TRAIN_UPDATE && c->repack_device && !c->train_copy_upload;
        if (c->debug_force_giveup && attempt == 0 && backward) {        // tests: the flag as a timed-out barrier would leave it, once
            HIPCHK(c, hipMemsetAsync(const_cast<unsigned*>(flag_dev), 1, sizeof(unsigned), c->stream));
            c->debug_force_giveup = false;
        }
        if (loss_out && !c->tr_loss_host) CHK(d2h(c, parts, c->tr_loss.p, (size_t)H * B * sizeof(double)));
        if (grad_out && backward) CHK(d2h(c, grad_out, c->tr_grad.p, (size_t)W_TOTAL * sizeof(float)));
        if (backward && !direct) CHK(d2h(c, gave_up, flag_dev, sizeof(unsigned)));
        bool repacked = false;
        if (mode == DRP_TRAIN_UPDATE) {
            const long iter = c->tr_iter + 1;
            const double bc1 = 1.0 - pow(c->tr_beta1, (double)iter), bc2 = 1.0 - pow(0.999, (double)iter);
            const float inf = __builtin_inff();
            if (direct) CHK(ensure_repack_maps(c));          // (allocates w_pin)
            hipLaunchKernelGGL(k_adam, dim3((W_TOTAL + 255) / 256), dim3(256), 0, c->stream, ptr<float>(c->w_raw),
                               ptr<float>(c->tr_grad), ptr<float>(c->tr_m), ptr<float>(c->tr_v), (int)W_TOTAL,
                               (float)(c->tr_lr / bc1), (float)sqrt(bc2), make_float4(-inf, -inf, -inf, -inf),
                               make_float4(inf, inf, inf, inf), (float)c->tr_beta1, direct ? c->w_pin : (float*)nullptr, flag_dev,
                               direct ? gave_up : (unsigned*)nullptr);
            if (hipGetLastError() != hipSuccess) { (void)drp_sync(c); return fail(c, DRP_EHIP, "k_adam launch"); }
            // the engines read packed copies of the weights: rebuild them from the blob (unchanged if the step was skipped)
            if (c->repack_device) {
                const int rc = repack_on_device(c, direct);
                if (rc != DRP_OK) { (void)drp_sync(c); return rc; }
                repacked = true;
            } else {
                std::vector<float> blob((size_t)W_TOTAL);
                CHK(d2h(c, blob.data(), c->w_raw.p, (size_t)W_TOTAL * sizeof(float)));
                CHK(guarded_wait(c, nullptr));
                CHK(install_weights(c, blob));
            }
        }
        CHK(drp_sync(c));
        if (repacked && !*gave_up) CHK(finish_repack(c));      // (a skipped step wrote no blob: the weights are what they were)
        if (!*gave_up) {
            if (mode == DRP_TRAIN_UPDATE) c->tr_iter += 1;
            break;
        }
        if (attempt > 0 || c->train_parts == 1)
            return fail(c, DRP_EHIP, "kmb_step_bwd: a workgroup waited two seconds for the others of its group, with one workgroup per group too");
        c->train_parts = 1;                     // the device is shared or masked: the groups' workgroups are not all resident
        c->dv(DV_TRAIN_BARRIER_RETRY);
    }
    if (loss_out) {
        double total = 0.0;                     // fixed order: step-major, then sample
        for (size_t q = 0; q < (size_t)H * B; ++q) total += parts[q];
        *loss_out = total;
    }
    return DRP_OK;
}

int drp_train_set_lr(drp_ctx* c, double lr) {
    if (!c || !c->tr_on) return fail(c, DRP_ESTATE, "drp_train_begin not called");
    if (!(lr > 0.0)) return fail(c, DRP_EINVAL, "bad lr %g", lr);
    c->tr_lr = lr;
    return DRP_OK;
}

int drp_get_weights(drp_ctx* c, float* blob_out, size_t n_floats) {
    CHK(need(c, true, false, false));
    if (!blob_out || n_floats != (size_t)W_TOTAL) return fail(c, DRP_EINVAL, "blob_out must hold %d floats", (int)W_TOTAL);
    memcpy(blob_out, c->w_host.data(), n_floats * sizeof(float));
    return DRP_OK;
}
