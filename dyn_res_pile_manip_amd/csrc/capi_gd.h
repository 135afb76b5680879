// capi_gd.h -- a section of the C ABI's translation unit (textually included by drp_capi.hip, in this order: capi_ctx.h,
// capi_pipeline.h, then inside extern "C": capi_core.h, capi_mpc.h, capi_prep.h, capi_gd.h, capi_train.h, capi_comm.h, capi_debug.h).
// Here: the gradient-descent planner (row f1): forward with tape, reverse mode, Adam, drp_gd_*.

// ---- gradient-descent planner (row f1) ----------------------------------------------------------
namespace {
// relation encoder backward on the matrix cores (kmb_edge_encode): one tile of 32 edge slots per wave, the tiles of a
// small batch spread one per CU
void launch_edge_encode_mfma(drp_ctx* c, const float* s_prev, int prev_mod, size_t prev_stride, int nb, const int16_t* idx,
                             const uint8_t* cnt, const float* gah, const unsigned* mht, size_t bn, int N, int B, float* gpos_edge,
                             const KbEdgeDump& dump) {
    const long ntiles = (long)B * (((long)N * DRP_K + 31) / 32);
    const unsigned grid = (unsigned)(ntiles < (long)c->n_cu ? ntiles : (long)c->n_cu);
    hipLaunchKernelGGL(kmb_edge_encode, dim3(grid), dim3(64 * MFMA_WAVES), KMB_EDGE_ENCODE_LDS, c->stream, ptr<float>(c->w_mfma),
                       ptr<float>(c->w_mfma_bwd), s_prev, prev_mod, prev_stride, ptr<float>(c->attr), nb, ptr<float>(c->dens), nb, idx,
                       cnt, gah, mht, bn, N, B, gpos_edge, dump);
}
int gd_forward_backward(drp_ctx* c) {
    const int nb = c->gd_nb, N = c->gd_N, B = c->gd_B, H = c->gd_H;
    const size_t bn = (size_t)B * N;
    const size_t hstride = (size_t)H * N * 3;
    hipStream_t st = c->stream;
    const bool rev_lds = N <= KB_REV_LDS_MAX_N && !c->rev_global_only;
    float* states = ptr<float>(c->states);
    float* eh = ptr<float>(c->eff_hist);
    unsigned* mh = ptr<unsigned>(c->tape_mask);
    auto d2d = [&](void* dst, const void* src, size_t bytes) -> int {
        HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
        return DRP_OK;
    };
    // ---- forward on the fused engine; km_prop<., TAPE> leaves what the backward pass needs: the
    //      effect after the encoder and after every propagation step, and the ReLU masks of the edges
    const int saved_engine = c->engine;
    c->engine = c->gd_engine;
    // the self-edge constants depend on attributes and densities only: computed once per GD problem,
    // again only if a rollout in between has reused the buffer
    int rc = DRP_OK;
    if (c->gd_cself_tag != c->cself_tag || c->gd_cself_tag == 0) {
        rc = prepare_cself(c, nb, N, B, &c->gd_cself, &c->gd_cself_ok);
        c->gd_cself_tag = c->cself_tag;
    }
    const float* cself = c->gd_cself;
    const uint8_t* cself_ok = c->gd_cself_ok;
    bool rev_built = false;
    for (int t = 0; t < H && rc == DRP_OK; ++t) {
        StepArgs a{};
        if (t == 0) { a.s_prev = ptr<float>(c->s_in); a.prev_mod = nb; a.prev_stride = (size_t)N * 3; }
        else { a.s_prev = states + (size_t)(t - 1) * N * 3; a.prev_mod = B; a.prev_stride = hstride; }
        a.attr = ptr<float>(c->attr); a.attr_mod = nb;
        a.dens = ptr<float>(c->dens); a.dens_mod = nb;
        a.actions = ptr<float>(c->actions) + (size_t)t * 4; a.act_stride = (size_t)H * 4;
        a.build_graph = true;
        a.s_out = states + (size_t)t * N * 3; a.out_stride = hstride;
        a.B = B; a.N = N;
        a.eff_hist = eh + (size_t)t * 4 * bn * 64;
        a.mask_hist = mh + (size_t)t * DRP_PSTEP * bn * DRP_K * 2;
        a.cself = cself; a.cself_ok = cself_ok;
        if (H == 1) { a.rev_off = ptr<int>(c->rev_off); a.rev = ptr<int>(c->rev); a.rev_built = &rev_built; }   // one set of reversed lists: the only step's
        // the step's impulses and neighbour lists are part of the tape: the step writes them there (its
        // workspace pointers are lent the tape's slices for the call) instead of being copied afterwards
        void* const save_sd = c->s_delta.p; void* const save_idx = c->nbr_idx.p; void* const save_cnt = c->nbr_cnt.p;
        c->s_delta.p = ptr<float>(c->tape_sdelta) + (size_t)t * bn * 3;
        c->nbr_idx.p = ptr<int16_t>(c->tape_idx) + (size_t)t * bn * DRP_K;
        c->nbr_cnt.p = ptr<uint8_t>(c->tape_cnt) + (size_t)t * bn;
        rc = run_step(c, a);
        c->s_delta.p = save_sd; c->nbr_idx.p = save_idx; c->nbr_cnt.p = save_cnt;
        if (rc != DRP_OK) break;
    }
    c->engine = saved_engine;
    CHK(rc);
    // reward of the final step only (planners.py:436-438) and its gradient, in one launch
    const float* vw = ptr<float>(c->w_valu);
    const float* wraw = ptr<float>(c->w_raw);
    float* g_state = ptr<float>(c->g_state);                 // [H][B,N,3]
    {
        ProbeScope ps(c, KC_BWD_REWARD);
        c->dv(DV_BWD_REWARD);
        hipLaunchKernelGGL(kb_reward, dim3(B), dim3(256), KB_REWARD_LDS(N), st, states + (size_t)(H - 1) * N * 3, hstride,
                           N, ptr<float>(c->goal_field), c->goal_h, c->goal_w, ptr<float>(c->goal_coor), c->goal_m, c->cam,
                           1, g_state + (size_t)(H - 1) * bn * 3, (size_t)N * 3, ptr<float>(c->rewards), c->gd_host_rewards);
    }
    for (int t = H - 1; t >= 0; --t) {
        const float* s_prev = (t == 0) ? ptr<float>(c->s_in) : states + (size_t)(t - 1) * N * 3;
        const int prev_mod = (t == 0) ? nb : B;
        const size_t prev_stride = (t == 0) ? (size_t)N * 3 : hstride;
        float* eht = eh + (size_t)t * 4 * bn * 64;
        const unsigned* mht = mh + (size_t)t * DRP_PSTEP * bn * DRP_K * 2;
        const int16_t* idx = ptr<int16_t>(c->tape_idx) + (size_t)t * bn * DRP_K;
        const uint8_t* cnt = ptr<uint8_t>(c->tape_cnt) + (size_t)t * bn;
        float* g_out = g_state + (size_t)t * bn * 3;
        float* gah = ptr<float>(c->g_agg_hist);
        if (!rev_built) {
            ProbeScope ps(c, KC_BWD_LISTS);
            c->dv(N <= 512 ? DV_REV_256 : DV_REV_1024);
            if (N <= 512)
                hipLaunchKernelGGL(kb_reverse_lists<256>, dim3(B), dim3(256), KB_REV_LDS(N, rev_lds), st, idx,
                                   cnt, N, ptr<int>(c->rev_off), ptr<int>(c->rev), rev_lds ? 1 : 0, (const int*)nullptr);
            else
                hipLaunchKernelGGL(kb_reverse_lists<1024>, dim3(B), dim3(1024), KB_REV_LDS(N, rev_lds), st, idx,
                                   cnt, N, ptr<int>(c->rev_off), ptr<int>(c->rev), rev_lds ? 1 : 0, (const int*)nullptr);
        }
        const int spw_b = (B + c->n_cu - 1) / c->n_cu;
        if (c->bwd_fused && c->bwd_rows && N <= KMB_ROWS_MAX) {
            // piles of up to 256 particles: a workgroup takes groups of whole samples with at most 256 rows, a wave keeps
            // its tile's rows in registers through all phases (kmb_rows_bwd).  Samples per group: the fewest that do not
            // add a round of groups over the CUs (fewer waves at work per CU, more CUs at work)
            const int g_max = KMB_ROWS_MAX / N;
            auto rounds = [&](int g) { return (((long)B + g - 1) / g + c->n_cu - 1) / c->n_cu; };
            int gps = g_max;
            while (gps > 1 && rounds(gps - 1) == rounds(g_max)) --gps;
            const long n_groups = ((long)B + gps - 1) / gps;
            ProbeScope ps(c, KC_BWD_NODE);
            c->dv(DV_BWD_ROWS);
            hipLaunchKernelGGL(kmb_rows_bwd, dim3((unsigned)(n_groups < (long)c->n_cu ? n_groups : (long)c->n_cu)), dim3(64 * KMB_FUSED_WAVES),
                               KMB_ROWS_LDS, st, ptr<float>(c->w_mfma), ptr<float>(c->w_mfma_bwd), ptr<uint16_t>(c->w_split6),
                               ptr<uint16_t>(c->w_split6_bwd), eht, mht, cnt, ptr<int>(c->rev_off),
                               ptr<int>(c->rev), g_out, (size_t)N * 3, ptr<float>(c->tape_sdelta) + (size_t)t * bn * 3, ptr<float>(c->attr),
                               nb, ptr<float>(c->dens), nb, N, B, gps, t > 0 ? gah : (float*)nullptr, ptr<float>(c->g_sdelta));
        } else if (c->bwd_fused && whole_samples(c, B, N) && ((long)spw_b * N + 31) / 32 >= c->bwd_fused_min_tiles) {
            // chip-filling batches: everything between the reward's gradient and the impulses' in one launch,
            // a workgroup owning whole samples (kmb_step_bwd)
            ProbeScope ps(c, KC_BWD_NODE);
            c->dv(DV_BWD_STEP);
            hipLaunchKernelGGL((kmb_step_bwd<false, false>), dim3((unsigned)((B + spw_b - 1) / spw_b)), dim3(64 * KMB_FUSED_WAVES), KMB_FUSED_LDS, st,
                               ptr<float>(c->w_mfma), ptr<float>(c->w_mfma_bwd), eht, mht, cnt, ptr<int>(c->rev_off), ptr<int>(c->rev),
                               g_out, (size_t)N * 3, ptr<float>(c->tape_sdelta) + (size_t)t * bn * 3, ptr<float>(c->attr), nb,
                               ptr<float>(c->dens), nb, N, B, spw_b, ptr<float>(c->g_eff), ptr<float>(c->g_cnode), gah,
                               ptr<float>(c->g_sdelta), KmbDump{}, 1, (unsigned*)nullptr, (unsigned*)nullptr);
        } else if ((long)B * ((N + 31) / 32) >= KMB_MIN_TILES && !c->bwd_valu_stages) {      // node stages on the matrix cores
            const float* mw = ptr<float>(c->w_mfma);
            const float* mb = ptr<float>(c->w_mfma_bwd);
            const long node_tiles = (long)B * ((N + 31) / 32);
            const dim3 ngrid(mfma_grid_spread(c, node_tiles)), nblk(64 * MFMA_WAVES);
            c->dv(DV_BWD_STAGES_MFMA);
            { ProbeScope ps(c, KC_BWD_NODE);
            hipLaunchKernelGGL(kmb_predict, ngrid, nblk, KMB_PREDICT_LDS, st, mw, mb, eht + 3 * bn * 64, g_out, (size_t)N * 3, N, B,
                               ptr<float>(c->g_eff), (float*)nullptr, (float*)nullptr);
            }
            // update of the last propagation step, then per step: edge terms, and in one launch the
            // projection of this step with the update of the one before
            { ProbeScope ps(c, KC_BWD_NODE);
            hipLaunchKernelGGL((kmb_node_step<false, true>), ngrid, nblk, KMB_STEP_LDS(false, true), st, mb, ptr<float>(c->g_eff),
                               ptr<float>(c->g_eff), (const float*)nullptr, eht + (size_t)DRP_PSTEP * bn * 64, ptr<float>(c->g_cnode), 1,
                               gah + (size_t)(DRP_PSTEP - 1) * bn * 64, N, B);
            }
            for (int p = DRP_PSTEP - 1; p >= 0; --p) {
                float* g_agg_p = gah + (size_t)p * bn * 64;
                const unsigned* mask_p = mht + (size_t)p * bn * DRP_K * 2;
                { ProbeScope ps(c, KC_BWD_EDGE);
                hipLaunchKernelGGL(kb_edge_terms, dim3(B), dim3(256), 0, st, g_agg_p, mask_p, cnt, ptr<int>(c->rev_off),
                                   ptr<int>(c->rev), N, ptr<float>(c->g_proj), 1);
                }
                if (p > 0)
                    { ProbeScope ps(c, KC_BWD_NODE);
                    hipLaunchKernelGGL((kmb_node_step<true, true>), ngrid, nblk, KMB_STEP_LDS(true, true), st, mb,
                                       ptr<float>(c->g_eff), ptr<float>(c->g_eff), ptr<float>(c->g_proj), eht + (size_t)p * bn * 64,
                                       ptr<float>(c->g_cnode), 0, gah + (size_t)(p - 1) * bn * 64, N, B);
                    }
                else
                    { ProbeScope ps(c, KC_BWD_NODE);
                    hipLaunchKernelGGL((kmb_node_step<true, false>), ngrid, nblk, KMB_STEP_LDS(true, false), st, mb,
                                       ptr<float>(c->g_eff), ptr<float>(c->g_eff), ptr<float>(c->g_proj), (const float*)nullptr, (float*)nullptr, 0,
                                       (float*)nullptr, N, B);
                    }
            }
            { ProbeScope ps(c, KC_BWD_NODE);
            hipLaunchKernelGGL(kmb_node_encode, ngrid, nblk, KMB_NODE_ENCODE_LDS, st, mw, mb,
                               ptr<float>(c->tape_sdelta) + (size_t)t * bn * 3, ptr<float>(c->attr), nb, ptr<float>(c->dens), nb,
                               eht, ptr<float>(c->g_eff), ptr<float>(c->g_cnode), N, B, ptr<float>(c->g_sdelta), (float*)nullptr,
                               (float*)nullptr, (float*)nullptr, (float*)nullptr);
            }
        } else {
            c->dv(DV_BWD_STAGES_VALU);
            { ProbeScope ps(c, KC_BWD_NODE);
            hipLaunchKernelGGL(kb_predict, dim3(B), dim3(256), 0, st, vw, wraw, eht + 3 * bn * 64, g_out, (size_t)N * 3, N,
                               ptr<float>(c->g_eff), (float*)nullptr, (float*)nullptr, 1);
            }
            for (int p = DRP_PSTEP - 1; p >= 0; --p) {
                float* g_agg_p = gah + (size_t)p * bn * 64;
                const unsigned* mask_p = mht + (size_t)p * bn * DRP_K * 2;
                { ProbeScope ps(c, KC_BWD_NODE);
                hipLaunchKernelGGL(kb_update, dim3(B), dim3(256), 0, st, wraw, eht + (size_t)(p + 1) * bn * 64,
                                   ptr<float>(c->g_eff), ptr<float>(c->g_cnode), p == DRP_PSTEP - 1 ? 1 : 0, N, g_agg_p, 1);
                }
                { ProbeScope ps(c, KC_BWD_EDGE);
                hipLaunchKernelGGL(kb_edge_terms, dim3(B), dim3(256), 0, st, g_agg_p, mask_p, cnt, ptr<int>(c->rev_off),
                                   ptr<int>(c->rev), N, ptr<float>(c->g_proj), 1);
                }
                { ProbeScope ps(c, KC_BWD_NODE);
                hipLaunchKernelGGL(kb_project, dim3(B), dim3(256), 0, st, wraw, ptr<float>(c->g_proj), N, ptr<float>(c->g_eff), 1);
                }
            }
            { ProbeScope ps(c, KC_BWD_NODE);
            hipLaunchKernelGGL(kb_node_encode, dim3(B), dim3(256), 0, st, vw, wraw,
                               ptr<float>(c->tape_sdelta) + (size_t)t * bn * 3, ptr<float>(c->attr), nb, ptr<float>(c->dens),
                               nb, eht, ptr<float>(c->g_eff), ptr<float>(c->g_cnode), N, ptr<float>(c->g_sdelta),
                               (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, 1);
            }
        }
        float* g_prev = nullptr;
        if (t > 0) {
            // d loss / d state[t-1] = residual share + relation encoder + gen_s_delta's position dependence
            g_prev = g_state + (size_t)(t - 1) * bn * 3;
            CHK(d2d(g_prev, g_out, bn * 3 * sizeof(float)));
            { ProbeScope ps(c, KC_BWD_EDGE);
            c->dv(c->bwd_edge_mfma ? DV_BWD_EDGE_MFMA : DV_BWD_EDGE_VALU);
            if (c->bwd_edge_mfma)
                launch_edge_encode_mfma(c, s_prev, prev_mod, prev_stride, nb, idx, cnt, gah, mht, bn, N, B, ptr<float>(c->gpos_edge), KbEdgeDump{});
            else
                hipLaunchKernelGGL(kb_edge_encode, dim3(B), dim3(256), KB_EDGE_ENCODE_LDS, st, vw, wraw,
                                   s_prev, prev_mod, prev_stride, ptr<float>(c->attr), nb, ptr<float>(c->dens), nb, idx, cnt,
                                   gah, mht, bn, N, g_prev, (size_t)N * 3, ptr<float>(c->gpos_edge), KbEdgeDump{}, 1);
            }
            { ProbeScope ps(c, KC_BWD_EDGE);
            hipLaunchKernelGGL(kb_gather_pos, dim3((N + 255) / 256, B), dim3(256), 0, st, ptr<float>(c->gpos_edge),
                               ptr<int>(c->rev_off), ptr<int>(c->rev), N, g_prev, (size_t)N * 3, c->bwd_edge_mfma ? 1 : 0, cnt);
            }
        }
        { ProbeScope ps(c, KC_BWD_PUSH);
        hipLaunchKernelGGL(kb_sdelta, dim3(B), dim3(256), 0, st, s_prev, prev_mod, prev_stride,
                           ptr<float>(c->actions) + (size_t)t * 4, (size_t)H * 4, ptr<float>(c->g_sdelta), N, c->cam,
                           ptr<float>(c->g_act) + (size_t)t * 4, (size_t)H * 4, g_prev, (size_t)N * 3, t == 0 ? c->gd_adam : KbAdam{});
        }
    }
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}
}  // namespace

int drp_gd_begin(drp_ctx* c, const float* s0, const float* attr, const float* dens, int nb, int N,
                 const float* actions, int B, int H, double lr, const float act_lo[4], const float act_hi[4]) {
    CHK(need(c, true, true, true));
    CHK(check_bn(c, B, N));
    if (!s0 || !attr || !dens || !actions || !act_lo || !act_hi) return fail(c, DRP_EINVAL, "null argument");
    if (H < 1 || H > 64) return fail(c, DRP_EINVAL, "bad horizon H=%d", H);
    if (nb <= 0 || B % nb != 0) return fail(c, DRP_EINVAL, "B must be a multiple of n_batch");
    HIPCHK(c, hipSetDevice(c->device));
    {
        // Adam moves the pushes, the clip keeps them in the box: bound by the box's diagonals and by the initial pushes
        const float box[8] = {act_lo[0], act_lo[1], act_hi[2], act_hi[3], act_hi[0], act_hi[1], act_lo[2], act_lo[3]};
        CHK(pick_tape_engine(c, max_abs(attr, (size_t)nb * N), max_abs(dens, (size_t)nb),
                             fmaxf(push_len_bound(c, box, 2), push_len_bound(c, actions, (size_t)B * H)), &c->gd_engine));
    }
    const size_t bn = (size_t)B * N;
    CHK(h2d(c, c->s_in, s0, (size_t)nb * N * 3 * sizeof(float)));
    CHK(h2d(c, c->attr, attr, (size_t)nb * N * sizeof(float)));
    CHK(h2d(c, c->dens, dens, (size_t)nb * sizeof(float)));
    CHK(h2d(c, c->actions, actions, (size_t)B * H * 4 * sizeof(float)));
    CHK(ensure_step_ws(c, B, N, c->gd_engine));
    CHK(ensure(c, c->states, (size_t)H * bn * 3 * sizeof(float)));
    CHK(ensure(c, c->rewards, (size_t)B * sizeof(float)));
    CHK(ensure(c, c->eff_hist, (size_t)H * 4 * bn * 64 * sizeof(float)));
    CHK(ensure(c, c->tape_sdelta, (size_t)H * bn * 3 * sizeof(float)));
    CHK(ensure(c, c->tape_idx, (size_t)H * bn * DRP_K * sizeof(int16_t)));
    CHK(ensure(c, c->tape_cnt, (size_t)H * bn));
    CHK(ensure(c, c->tape_mask, (size_t)H * DRP_PSTEP * bn * DRP_K * 2 * sizeof(unsigned)));
    CHK(ensure(c, c->g_agg_hist, (size_t)DRP_PSTEP * bn * 64 * sizeof(float)));
    CHK(ensure(c, c->rev_off, (size_t)B * (N + 1) * sizeof(int)));
    CHK(ensure(c, c->rev, bn * DRP_K * sizeof(int)));
    CHK(ensure(c, c->gpos_edge, bn * DRP_K * 4 * sizeof(float)));
    CHK(ensure(c, c->g_eff, bn * 64 * sizeof(float)));
    CHK(ensure(c, c->g_cnode, bn * 64 * sizeof(float)));
    CHK(ensure(c, c->g_agg, bn * 64 * sizeof(float)));
    CHK(ensure(c, c->g_proj, bn * 128 * sizeof(float)));
    CHK(ensure(c, c->g_state, (size_t)H * bn * 3 * sizeof(float)));
    CHK(ensure(c, c->g_sdelta, bn * 3 * sizeof(float)));
    CHK(ensure(c, c->g_act, (size_t)B * H * 4 * sizeof(float)));
    CHK(ensure(c, c->adam_m, (size_t)B * H * 4 * sizeof(float)));
    CHK(ensure(c, c->adam_v, (size_t)B * H * 4 * sizeof(float)));
    HIPCHK(c, hipMemsetAsync(c->adam_m.p, 0, (size_t)B * H * 4 * sizeof(float), c->stream));
    HIPCHK(c, hipMemsetAsync(c->adam_v.p, 0, (size_t)B * H * 4 * sizeof(float), c->stream));
    CHK(guarded_wait(c, nullptr));
    c->gd_nb = nb; c->gd_N = N; c->gd_B = B; c->gd_H = H; c->gd_iter = 0; c->gd_lr = lr;
    for (int q = 0; q < DRP_GD_SLOTS; ++q) c->gd_pending[q] = false;           // a new problem drops what the last one left in flight
    c->gd_cself_tag = 0;
    memcpy(c->gd_lo, act_lo, 4 * sizeof(float));
    memcpy(c->gd_hi, act_hi, 4 * sizeof(float));
    c->lastH = H;
    c->gd_on = true;
    c->mpc_on = false;
    return DRP_OK;
}

int drp_gd_grad(drp_ctx* c, float* rewards_out, float* grad_act_out, float* grad_state_out) {
    if (!c || !c->gd_on) return fail(c, DRP_ESTATE, "drp_gd_begin not called");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(gd_forward_backward(c));
    const size_t bn = (size_t)c->gd_B * c->gd_N;
    if (rewards_out) CHK(d2h(c, rewards_out, c->rewards.p, (size_t)c->gd_B * sizeof(float)));
    if (grad_act_out) CHK(d2h(c, grad_act_out, c->g_act.p, (size_t)c->gd_B * c->gd_H * 4 * sizeof(float)));
    if (grad_state_out) {
        // device layout [H][B,N,3] -> caller layout [B,H,N,3]
        const size_t row = (size_t)c->gd_N * 3 * sizeof(float);
        for (int t = 0; t < c->gd_H; ++t)
            HIPCHK(c, hipMemcpy2DAsync(grad_state_out + (size_t)t * c->gd_N * 3, (size_t)c->gd_H * row,
                                       ptr<float>(c->g_state) + (size_t)t * bn * 3, row, row, c->gd_B,
                                       hipMemcpyDeviceToHost, c->stream));
    }
    return drp_sync(c);
}

namespace {
// one iteration on the stream: forward, backward, Adam, clip -- the optimiser step of a row in the kb_sdelta launch that
// completes the row's gradient (rollout step 0's, the last of the backward pass): one launch fewer per iteration
int gd_iteration(drp_ctx* c) {
    // torch.optim.Adam: step_size = lr / (1 - beta1^t), denom = sqrt(v) / sqrt(1 - beta2^t) + eps
    const double it = (double)(c->gd_iter + 1);
    const double bc1 = 1.0 - pow(0.9, it), bc2 = 1.0 - pow(0.999, it);
    KbAdam a{};
    a.act = ptr<float>(c->actions); a.m = ptr<float>(c->adam_m); a.v = ptr<float>(c->adam_v); a.act_copy = c->gd_host_actions;
    a.n_row = c->gd_H * 4;
    a.step_size = (float)(c->gd_lr / bc1); a.bc2_sqrt = (float)sqrt(bc2); a.b1 = 0.9f;
    a.lo = make_float4(c->gd_lo[0], c->gd_lo[1], c->gd_lo[2], c->gd_lo[3]);
    a.hi = make_float4(c->gd_hi[0], c->gd_hi[1], c->gd_hi[2], c->gd_hi[3]);
    c->gd_adam = a;
    const int rc = gd_forward_backward(c);
    c->gd_adam = KbAdam{};
    CHK(rc);
    c->gd_iter += 1;
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}
}  // namespace

int drp_gd_step(drp_ctx* c, float* rewards_out) {
    if (!c || !c->gd_on) return fail(c, DRP_ESTATE, "drp_gd_begin not called");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(gd_iteration(c));
    if (rewards_out) {
        CHK(d2h(c, rewards_out, c->rewards.p, (size_t)c->gd_B * sizeof(float)));
        return drp_sync(c);
    }
    return DRP_OK;
}

// The planner's loop needs every iteration's rewards and updated pushes on the host (per-column bookkeeping,
// planners.py:721-727), but no iteration waits for the host: slot s of two takes the iteration's results into pinned
// memory behind the kernels, the caller enqueues the NEXT iteration before it waits for this one.
int drp_gd_step_async(drp_ctx* c, int slot) {
    if (!c || !c->gd_on) return fail(c, DRP_ESTATE, "drp_gd_begin not called");
    if (slot < 0 || slot >= DRP_GD_SLOTS) return fail(c, DRP_EINVAL, "slot must be 0 .. %d", DRP_GD_SLOTS - 1);
    if (c->gd_pending[slot]) return fail(c, DRP_ESTATE, "slot %d holds an iteration nobody has waited for", slot);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nr = (size_t)c->gd_B, na = (size_t)c->gd_B * c->gd_H * 4;
    if (c->gd_pin_floats < nr + na) {
        for (int q = 0; q < DRP_GD_SLOTS; ++q) {
            if (c->gd_pending[q]) return fail(c, DRP_ESTATE, "the batch grew while an iteration was in flight");
            if (c->gd_pin[q]) HIPCHK(c, hipHostFree(c->gd_pin[q]));
            c->gd_pin[q] = nullptr;
            HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->gd_pin[q]), (nr + na) * sizeof(float), hipHostMallocDefault));
            if (!c->gd_ev[q]) HIPCHK(c, hipEventCreateWithFlags(&c->gd_ev[q], hipEventDisableTiming));
        }
        c->gd_pin_floats = nr + na;
    }
    // the iteration's own kernels write the slot (pinned host memory is device-visible): kb_reward the rewards, k_adam
    // the updated pushes -- two copies fewer on the stream per iteration (they were 27 of 197 us at 20 particles)
    c->gd_host_rewards = c->gd_pin[slot];
    c->gd_host_actions = c->gd_pin[slot] + nr;
    const int rc_it = gd_iteration(c);
    c->gd_host_rewards = c->gd_host_actions = nullptr;
    CHK(rc_it);
    HIPCHK(c, hipEventRecord(c->gd_ev[slot], c->stream));
    c->gd_pending[slot] = true;
    return DRP_OK;
}

int drp_gd_wait(drp_ctx* c, int slot, float* rewards_out, float* actions_out) {
    if (!c) return DRP_EINVAL;
    if (slot < 0 || slot >= DRP_GD_SLOTS || !c->gd_pending[slot]) return fail(c, DRP_ESTATE, "no iteration in flight in slot %d", slot);
    HIPCHK(c, hipSetDevice(c->device));
    c->gd_pending[slot] = false;
    CHK(guarded_wait(c, c->gd_ev[slot]));
    const size_t nr = (size_t)c->gd_B, na = (size_t)c->gd_B * c->gd_H * 4;
    if (rewards_out) memcpy(rewards_out, c->gd_pin[slot], nr * sizeof(float));
    if (actions_out) memcpy(actions_out, c->gd_pin[slot] + nr, na * sizeof(float));
    return DRP_OK;
}

int drp_gd_get(drp_ctx* c, float* actions_out) {
    if (!c || !c->gd_on) return fail(c, DRP_ESTATE, "drp_gd_begin not called");
    if (!actions_out) return fail(c, DRP_EINVAL, "null buffer");
    CHK(d2h(c, actions_out, c->actions.p, (size_t)c->gd_B * c->gd_H * 4 * sizeof(float)));
    return drp_sync(c);
}
