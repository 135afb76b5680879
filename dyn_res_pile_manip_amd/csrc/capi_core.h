// capi_core.h -- a section of the C ABI's translation unit (textually included by drp_capi.hip, in this order: capi_ctx.h,
// capi_pipeline.h, then inside extern "C": capi_core.h, capi_mpc.h, capi_prep.h, capi_gd.h, capi_train.h, capi_comm.h, capi_debug.h).
// Here: life cycle, model constants and the single operations on host buffers (drp_create ... drp_reward).

int drp_create(int device, drp_ctx** out) {
    if (!out) return fail(nullptr, DRP_EINVAL, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, DRP_EHIP, "no HIP device available: %s", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(nullptr, DRP_EINVAL, "device %d out of range (%d)", device, n);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, DRP_EHIP, "hipSetDevice: %s", hipGetErrorString(e));
    drp_ctx* c = new drp_ctx();
    c->device = device;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(nullptr, DRP_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        c->n_cu = prop.multiProcessorCount;
    c->self_const = getenv("DRP_NO_SELF_CONST") == nullptr;
    c->prop3 = getenv("DRP_NO_PROP3") == nullptr;
    c->graph_strips = getenv("DRP_NO_GRAPH_STRIPS") == nullptr;
    c->graph_cells = getenv("DRP_NO_GRAPH_CELLS") == nullptr;
    if (const char* e = getenv("DRP_GRAPH_CELLS_MIN_N")) c->graph_cells_min_n = atoi(e);
    if (const char* e = getenv("DRP_GRAPH_CELLS_HB")) c->graph_cells_hb = (float)atof(e);
    if (const char* e = getenv("DRP_GRAPH_CELLS_HALO")) c->graph_cells_halo = (float)atof(e);
    c->rollout_fused = getenv("DRP_NO_ROLLOUT_FUSED") == nullptr;
    c->repack_device = getenv("DRP_NO_REPACK_DEVICE") == nullptr;
    c->bwd_edge_mfma = getenv("DRP_NO_BWD_EDGE_MFMA") == nullptr;
    c->prop_spread = getenv("DRP_NO_PROP_SPREAD") == nullptr;
    c->wgrad_mfma = getenv("DRP_NO_WGRAD_MFMA") == nullptr;
    c->wgrad_defer = getenv("DRP_NO_WGRAD_DEFER") == nullptr;
    if (const char* e = getenv("DRP_GRAPH_Q4")) c->graph_q4 = atoi(e);
    if (const char* e = getenv("DRP_ROLLOUT_MAX_N")) { c->rollout_max_n = atoi(e); c->rollout_mid_n = 0; c->rollout_max_rows = KM_ROLLOUT_MAX_ROWS; }
    if (const char* e = getenv("DRP_PROP_PAIR_ROWS")) c->prop_pair_rows = std::min(256, std::max(0, atoi(e)));   // km_rollout<pair> keeps 16 B per row in the 4 KB behind the encoder's matrices
    if (const char* e = getenv("DRP_PROP_PAIR_ALWAYS")) c->prop_pair_always = std::max(0, atoi(e));
    if (const char* e = getenv("DRP_PROP_PAIR_DEG10")) c->prop_pair_deg10 = std::max(0, atoi(e));
    c->bwd_fused = getenv("DRP_NO_BWD_FUSED") == nullptr;
    c->bwd_rows = getenv("DRP_NO_BWD_ROWS") == nullptr;
    c->bwd_valu_stages = getenv("DRP_BWD_VALU_STAGES") != nullptr;
    if (const char* e = getenv("DRP_TRAIN_PARTS")) c->train_parts = atoi(e);
    c->debug_force_giveup = getenv("DRP_DEBUG_FORCE_GIVEUP") != nullptr;
    c->train_copy_upload = getenv("DRP_TRAIN_COPY_UPLOAD") != nullptr;
    if (const char* e = getenv("DRP_TRAIN_COOP")) c->train_coop = atoi(e);
    c->graph_encode = getenv("DRP_NO_GRAPH_ENCODE") == nullptr;
    if (const char* e = getenv("DRP_TRAIN_FUSED")) c->train_fused = atoi(e);
    c->graph_rev = getenv("DRP_NO_GRAPH_REV") == nullptr;
    c->comm_always = getenv("DRP_COMM_ALWAYS") != nullptr;
    if (const char* e = getenv("DRP_COMM_TIMEOUT_S")) { const double v = atof(e); if (v > 0.0) c->comm_timeout_s = v; }
    if (const char* e = getenv("DRP_COMM_INIT_TIMEOUT_S")) { const double v = atof(e); if (v > 0.0) c->comm_init_timeout_s = v; }
    c->rev_global_only = getenv("DRP_REV_GLOBAL") != nullptr;
    if (const char* e = getenv("DRP_ECACHE_MAX_MB")) c->ecache_max_mb = std::max(0, atoi(e));
    if (const char* e = getenv("DRP_ECACHE_MAX_N")) { c->ecache_max_n = std::max(0, atoi(e)); c->ecache_full_n = 257; }
    if (const char* e = getenv("DRP_ECACHE_TAPE_MAX_N")) c->ecache_tape_max_n = std::max(0, atoi(e));
    if (hipFuncSetAttribute((const void*)k_graph, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_graph_q4, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_graph_q4_encode, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_elite_local, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_elite_update, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_graph_strips_q<GRAPH_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_graph_strips_q<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)kb_reward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KB_REWARD_LDS(4096)) != hipSuccess ||
        hipFuncSetAttribute((const void*)kb_edge_encode, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KB_EDGE_ENCODE_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)kb_reverse_lists<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KB_REV_LDS(KB_REV_LDS_MAX_N, 1)) != hipSuccess ||
        hipFuncSetAttribute((const void*)kb_reverse_lists<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KB_REV_LDS(KB_REV_LDS_MAX_N, 1)) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_aggregate_lds, hipFuncAttributeMaxDynamicSharedMemorySize,
                            K_AGG_LDS_MAX_N * 256) != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(nullptr, DRP_EHIP, "hipFuncSetAttribute (dynamic LDS size of k_graph, kb_edge_encode, kb_reverse_lists or k_aggregate_lds) failed");
    }
    // the MFMA kernels keep packed weights + per-wave transposition tiles in LDS (> 64 KiB)
    if (hipFuncSetAttribute((const void*)km_edge_encode, hipFuncAttributeMaxDynamicSharedMemorySize, KM_EDGE_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_node_encode, hipFuncAttributeMaxDynamicSharedMemorySize, KM_NODE_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_edge_encode_split, hipFuncAttributeMaxDynamicSharedMemorySize, KM_EDGE_SPLIT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_node_encode_split, hipFuncAttributeMaxDynamicSharedMemorySize, KM_NODE_SPLIT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<false, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(false)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(false)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<false, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(false)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(false)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<false, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(false)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(false)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<false, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(false)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<false, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(false)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<true, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(true)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(true)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<true, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(true)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<true, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(true)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<true, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(true)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(true)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<true, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(true)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop<true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP_LDS(true)) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, false, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, false, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, false, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, false, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, false, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, true, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, true, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, true, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, true, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<false, true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, false, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, false, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, false, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, false, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, false, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, true, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, true, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, true, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, true, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_prop3<true, true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_PROP3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<false, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<false, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<false, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<false, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<false, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<true, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<true, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<true, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<true, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_rollout<true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_ROLLOUT_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)kmb_step_bwd<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KMB_FUSED_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)kmb_step_bwd<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, KMB_FUSED_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)kmb_step_bwd<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, KMB_COOP_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)kmb_rows_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, KMB_ROWS_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)kmb_edge_encode, hipFuncAttributeMaxDynamicSharedMemorySize, KMB_EDGE_ENCODE_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)kt_wgrad_multi, hipFuncAttributeMaxDynamicSharedMemorySize, KT_WGRAD_MULTI_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)kt_wgrad_mfma_multi, hipFuncAttributeMaxDynamicSharedMemorySize, KT_WGRAD_MULTI_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_update<false>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_UPD_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)km_update<true>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_UPD_LDS) != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(nullptr, DRP_EHIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    }
    c->engine = DRP_ENGINE_FUSED;
    *out = c;
    return DRP_OK;
}

void drp_destroy(drp_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)guarded_wait(c, nullptr);           // a collective that cannot finish must not keep the destructor
    helpers_wait(5.0, c);                     // no helper thread of this context (an abort, an init) inside RCCL while its stream goes away
    if (c->comm) { RcclApi* R = rccl_api(); if (R) (void)R->CommDestroy(c->comm); c->comm = nullptr; }
    DevBuf* bufs[] = {&c->probe_work, &c->ecache, &c->tape_mask, &c->g_agg_hist, &c->rev_off, &c->rev, &c->gpos_edge, &c->tape_sdelta, &c->tape_idx, &c->tape_cnt, &c->eff_hist, &c->g_eff, &c->g_cnode, &c->g_agg, &c->g_proj, &c->g_state,
                      &c->g_sdelta, &c->g_act, &c->adam_m, &c->adam_v, &c->w_raw, &c->w_valu, &c->w_mfma, &c->w_mfma_bwd, &c->w_split, &c->w_split6, &c->w_split6_bwd, &c->proj2, &c->goal_field, &c->goal_coor, &c->s_in,
                      &c->attr, &c->dens, &c->s_delta, &c->nbr_idx, &c->nbr_cnt, &c->eff, &c->c_node,
                      &c->agg, &c->proj, &c->c_edge, &c->states, &c->actions, &c->rewards, &c->s_out,
                      &c->scratch, &c->nominal, &c->noise, &c->partials, &c->gathered, &c->stats, &c->elite, &c->elite_all, &c->xchg, &c->cself,
                      &c->px_depth, &c->px_mask, &c->px_blk, &c->px_bmin, &c->px_bmax, &c->px_grid, &c->px_pcd, &c->px_keys,
                      &c->px_cellcnt, &c->px_cellfill, &c->px_celloff, &c->px_list, &c->px_down, &c->px_down32, &c->px_init,
                      &c->px_dist, &c->px_chosen, &c->px_pts, &c->px_r, &c->px_rr, &c->px_out,
                      &c->gl_goal, &c->gl_seg, &c->gl_tmp, &c->gl_dist, &c->gl_blk, &c->gl_pix, &c->gl_fps,
                      &c->tr_part, &c->tr_arena, &c->re_shift_dev, &c->tr_grad, &c->tr_m, &c->tr_v, &c->tr_loss, &c->agg_hist,
                      &c->tr_hact, &c->tr_gh, &c->tr_gpe, &c->tr_a1n, &c->tr_gh1, &c->tr_xn, &c->ed_re, &c->ed_a2, &c->ed_a1,
                      &c->ed_x0, &c->ed_gce, &c->ed_g3, &c->ed_g2, &c->ed_g1, &c->roll_args, &c->map_valu, &c->map_mfma, &c->map_mfma_bwd,
                      &c->wg_jobs_dev, &c->wg_idx_dev};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (hipEvent_t ev : c->probe_ev) (void)hipEventDestroy(ev);
    for (int q = 0; q < DRP_GD_SLOTS; ++q) {
        if (c->gd_pin[q]) (void)hipHostFree(c->gd_pin[q]);
        if (c->gd_ev[q]) (void)hipEventDestroy(c->gd_ev[q]);
    }
    for (int q = 0; q < 2; ++q) {
        if (c->mpc_pin[q]) (void)hipHostFree(c->mpc_pin[q]);
        if (c->mpc_ev[q]) (void)hipEventDestroy(c->mpc_ev[q]);
    }
    if (c->w_pin) (void)hipHostFree(c->w_pin);
    if (c->tr_pin) (void)hipHostFree(c->tr_pin);
    if (c->deg_stat) (void)hipHostFree(c->deg_stat);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* drp_last_error(const drp_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int drp_sync(drp_ctx* c) {
    if (!c) return DRP_EINVAL;
    return guarded_wait(c, nullptr);
}

int drp_set_engine(drp_ctx* c, int engine) {
    if (!c) return DRP_EINVAL;
    if (engine == DRP_ENGINE_VALU) { c->engine = engine; return DRP_OK; }
    if (engine == DRP_ENGINE_MFMA || engine == DRP_ENGINE_SPLIT || engine == DRP_ENGINE_FUSED) {
        c->engine = engine;
        return DRP_OK;
    }
    return fail(c, DRP_EINVAL, "engine %d not available in this build", engine);
}

int drp_device_info(drp_ctx* c, char* name, size_t name_len, int* n_cu, size_t* hbm_bytes) {
    if (!c) return DRP_EINVAL;
    hipDeviceProp_t p;
    HIPCHK(c, hipGetDeviceProperties(&p, c->device));
    if (name && name_len) snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    return DRP_OK;
}

int drp_load_weights(drp_ctx* c, const float* blob, size_t n_floats, float adj_thresh) {
    if (!c || !blob) return DRP_EINVAL;
    if (n_floats != (size_t)W_TOTAL)
        return fail(c, DRP_EINVAL, "weight blob has %zu floats, expected %d", n_floats, (int)W_TOTAL);
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<float> v;
    pack_valu(blob, v);
    CHK(h2d(c, c->w_raw, blob, n_floats * sizeof(float)));
    CHK(h2d(c, c->w_valu, v.data(), v.size() * sizeof(float)));
    {
        std::vector<float> m;
        pack_mfma(blob, m);
        CHK(h2d(c, c->w_mfma, m.data(), m.size() * sizeof(float)));
        std::vector<float> mbv;
        pack_mfma_bwd(blob, mbv);
        CHK(h2d(c, c->w_mfma_bwd, mbv.data(), mbv.size() * sizeof(float)));
        CHK(guarded_wait(c, nullptr));     // mbv is about to go out of scope... kept alive until here
        set_split_range(c, blob);
        std::vector<uint16_t> sp;
        pack_split(blob, sp, c->re_range.shift);
        CHK(h2d(c, c->w_split, sp.data(), sp.size() * sizeof(uint16_t)));
        std::vector<uint16_t> sp6;
        pack_split6(blob, sp6);
        CHK(h2d(c, c->w_split6, sp6.data(), sp6.size() * sizeof(uint16_t)));
        // the transposed layers of the GD planner's backward pass in the same split: packed on the device from the raw blob
        CHK(ensure(c, c->w_split6_bwd, (size_t)SB6_TOTAL * 16));
        hipLaunchKernelGGL(kt_repack_split6_bwd, dim3(6 * 16), dim3(256), 0, c->stream, ptr<float>(c->w_raw), ptr<uint16_t>(c->w_split6_bwd));
        CHK(guarded_wait(c, nullptr));     // sp6 too
        CHK(guarded_wait(c, nullptr));     // m, sp are about to go out of scope
    }
    CHK(guarded_wait(c, nullptr));
    c->w_host.assign(blob, blob + n_floats);
    c->adj_thresh = adj_thresh;
    // threshold = adj_thresh * adj_thresh in Python doubles, then an fp32 scalar
    // (model/gnn_dyn.py:229,236)
    c->thr = (float)((double)adj_thresh * (double)adj_thresh);
    c->have_weights = true;
    return DRP_OK;
}

int drp_set_camera(drp_ctx* c, const float m34[12], float global_scale, const float intr[4]) {
    if (!c || !m34 || !intr) return DRP_EINVAL;
    memcpy(c->cam.m, m34, 12 * sizeof(float));
    c->cam.gs = global_scale;
    c->cam.fx = intr[0]; c->cam.fy = intr[1]; c->cam.cx = intr[2]; c->cam.cy = intr[3];
    c->have_cam = true;
    return DRP_OK;
}

int drp_set_goal(drp_ctx* c, const float* field, int h, int w, const float* goal_coor, int m) {
    if (!c || !field || !goal_coor || h <= 0 || w <= 0 || m <= 0) return fail(c, DRP_EINVAL, "bad goal");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(h2d(c, c->goal_field, field, (size_t)h * w * sizeof(float)));
    CHK(h2d(c, c->goal_coor, goal_coor, (size_t)m * 2 * sizeof(float)));
    CHK(guarded_wait(c, nullptr));
    c->goal_h = h; c->goal_w = w; c->goal_m = m;
    c->have_goal = true;
    return DRP_OK;
}

int drp_gen_s_delta(drp_ctx* c, const float* s_cur, const float* action, int B, int N, float* out) {
    CHK(need(c, false, true, false));
    CHK(check_bn(c, B, N));
    if (!s_cur || !action || !out) return fail(c, DRP_EINVAL, "null buffer");
    HIPCHK(c, hipSetDevice(c->device));
    end_sessions(c);
    CHK(h2d(c, c->s_in, s_cur, (size_t)B * N * 3 * sizeof(float)));
    CHK(h2d(c, c->actions, action, (size_t)B * 4 * sizeof(float)));
    CHK(ensure(c, c->s_delta, (size_t)B * N * 3 * sizeof(float)));
    hipLaunchKernelGGL(k_sdelta, dim3(B), dim3(256), 0, c->stream, ptr<float>(c->s_in),
                       ptr<float>(c->actions), N, ptr<float>(c->s_delta), c->cam);
    HIPCHK(c, hipGetLastError());
    CHK(d2h(c, out, c->s_delta.p, (size_t)B * N * 3 * sizeof(float)));
    return drp_sync(c);
}

int drp_build_graph(drp_ctx* c, const float* s_cur, const float* s_delta, int B, int N,
                    int16_t* nbr_idx_out, uint8_t* nbr_cnt_out) {
    CHK(need(c, true, false, false));
    CHK(check_bn(c, B, N));
    if (!s_cur || !s_delta || !nbr_idx_out || !nbr_cnt_out) return fail(c, DRP_EINVAL, "null buffer");
    HIPCHK(c, hipSetDevice(c->device));
    end_sessions(c);
    CHK(ensure_step_ws(c, B, N));
    CHK(h2d(c, c->s_in, s_cur, (size_t)B * N * 3 * sizeof(float)));
    CHK(h2d(c, c->s_delta, s_delta, (size_t)B * N * 3 * sizeof(float)));
    {
    ProbeScope ps(c, KC_GRAPH);
    launch_graph(c, c->stream, ptr<float>(c->s_in), B, (size_t)N * 3, (const float*)nullptr, (size_t)0,
                 ptr<float>(c->s_delta), B, N, ptr<int16_t>(c->nbr_idx), ptr<uint8_t>(c->nbr_cnt), 0, false);
    }
    HIPCHK(c, hipGetLastError());
    CHK(d2h(c, nbr_idx_out, c->nbr_idx.p, (size_t)B * N * DRP_K * sizeof(int16_t)));
    CHK(d2h(c, nbr_cnt_out, c->nbr_cnt.p, (size_t)B * N));
    return drp_sync(c);
}

static int step_common(drp_ctx* c, const float* a_cur, const float* s_cur, const float* s_delta,
                       const float* dens, const int16_t* nbr_idx, const uint8_t* nbr_cnt, int B, int N,
                       float* s_pred_out) {
    CHK(need(c, true, false, false));
    CHK(check_bn(c, B, N));
    if (!a_cur || !s_cur || !s_delta || !dens || !s_pred_out) return fail(c, DRP_EINVAL, "null buffer");
    HIPCHK(c, hipSetDevice(c->device));
    end_sessions(c);
    CHK(range_check(c, max_abs(a_cur, (size_t)B * N), max_abs(dens, (size_t)B), max_abs(s_delta, (size_t)B * N * 3)));
    CHK(ensure_step_ws(c, B, N));
    const size_t bn = (size_t)B * N;
    CHK(h2d(c, c->s_in, s_cur, bn * 3 * sizeof(float)));
    CHK(h2d(c, c->s_delta, s_delta, bn * 3 * sizeof(float)));
    CHK(h2d(c, c->attr, a_cur, bn * sizeof(float)));
    CHK(h2d(c, c->dens, dens, (size_t)B * sizeof(float)));
    CHK(ensure(c, c->s_out, bn * 3 * sizeof(float)));
    if (nbr_idx) {
        CHK(h2d(c, c->nbr_idx, nbr_idx, bn * DRP_K * sizeof(int16_t)));
        CHK(h2d(c, c->nbr_cnt, nbr_cnt, bn));
    }
    StepArgs a{};
    a.s_prev = ptr<float>(c->s_in); a.prev_mod = B; a.prev_stride = (size_t)N * 3;
    a.attr = ptr<float>(c->attr); a.attr_mod = B;
    a.dens = ptr<float>(c->dens); a.dens_mod = B;
    a.actions = nullptr; a.act_stride = 0;
    a.build_graph = (nbr_idx == nullptr);
    a.s_out = ptr<float>(c->s_out); a.out_stride = (size_t)N * 3;
    a.B = B; a.N = N;
    CHK(run_step(c, a));
    CHK(d2h(c, s_pred_out, c->s_out.p, bn * 3 * sizeof(float)));
    return drp_sync(c);
}

int drp_step(drp_ctx* c, const float* a_cur, const float* s_cur, const float* s_delta,
             const float* dens, int B, int N, float* s_pred_out) {
    return step_common(c, a_cur, s_cur, s_delta, dens, nullptr, nullptr, B, N, s_pred_out);
}

int drp_forward(drp_ctx* c, const float* a_cur, const float* s_cur, const float* s_delta,
                const float* dens, const int16_t* nbr_idx, const uint8_t* nbr_cnt, int B, int N,
                float* s_pred_out) {
    if (!nbr_idx || !nbr_cnt) return fail(c, DRP_EINVAL, "null neighbour lists");
    return step_common(c, a_cur, s_cur, s_delta, dens, nbr_idx, nbr_cnt, B, N, s_pred_out);
}

int drp_rollout(drp_ctx* c, const float* s0, const float* attr, const float* dens, int nb, int N,
                const float* actions, int B, int H, float* states_out, float* reward_out) {
    CHK(need(c, true, true, reward_out != nullptr));
    CHK(check_bn(c, B, N));
    if (!s0 || !attr || !dens || !actions) return fail(c, DRP_EINVAL, "null buffer");
    if (nb <= 0 || H <= 0 || B % nb != 0)
        return fail(c, DRP_EINVAL, "bad rollout shape nb=%d B=%d H=%d (B must be a multiple of nb)", nb, B, H);
    HIPCHK(c, hipSetDevice(c->device));
    end_sessions(c);
    CHK(range_check(c, max_abs(attr, (size_t)nb * N), max_abs(dens, (size_t)nb), push_len_bound(c, actions, (size_t)B * H)));
    CHK(h2d(c, c->s_in, s0, (size_t)nb * N * 3 * sizeof(float)));
    CHK(h2d(c, c->attr, attr, (size_t)nb * N * sizeof(float)));
    CHK(h2d(c, c->dens, dens, (size_t)nb * sizeof(float)));
    CHK(h2d(c, c->actions, actions, (size_t)B * H * 4 * sizeof(float)));
    CHK(run_rollout(c, nb, N, B, H, reward_out != nullptr, false));
    if (states_out) CHK(d2h(c, states_out, c->states.p, (size_t)B * H * N * 3 * sizeof(float)));
    if (reward_out) CHK(d2h(c, reward_out, c->rewards.p, (size_t)B * H * sizeof(float)));
    return drp_sync(c);
}

int drp_reward(drp_ctx* c, const float* state, int Bp, int N, int normalize, float* reward_out) {
    CHK(need(c, false, true, true));
    CHK(check_bn(c, Bp, N));
    if (!state || !reward_out) return fail(c, DRP_EINVAL, "null buffer");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(h2d(c, c->s_out, state, (size_t)Bp * N * 3 * sizeof(float)));
    CHK(ensure(c, c->scratch, (size_t)Bp * sizeof(float)));
    CHK(run_reward(c, ptr<float>(c->s_out), (size_t)N * 3, Bp, N, normalize, ptr<float>(c->scratch)));
    CHK(d2h(c, reward_out, c->scratch.p, (size_t)Bp * sizeof(float)));
    return drp_sync(c);
}
