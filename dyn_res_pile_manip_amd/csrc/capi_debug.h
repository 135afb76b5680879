// capi_debug.h -- a section of the C ABI's translation unit (textually included by drp_capi.hip, in this order: capi_ctx.h,
// capi_pipeline.h, then inside extern "C": capi_core.h, capi_mpc.h, capi_prep.h, capi_gd.h, capi_train.h, capi_comm.h, capi_debug.h).
// Here: measurement and debugging entry points (probes, dispatch introspection, range info, buffer fetch).

// ---- measurement / debugging -----------------------------------------------------------------
#ifdef PROP_STAMPS
int drp_debug_prop_stamps(drp_ctx* c, unsigned long long* out8, int reset) {
    (void)c;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    std::vector<unsigned long long> h(4096 * 8);
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_prop_stamps), h.size() * 8) != hipSuccess) return -1;
    if (out8) {
        for (int q = 0; q < 8; ++q) out8[q] = 0;
        for (size_t i = 0; i < h.size(); ++i) out8[i & 7] += h[i];
    }
    if (reset) {
        std::fill(h.begin(), h.end(), 0ull);
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_prop_stamps), h.data(), h.size() * 8) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#ifdef PROP_STAMPS
int drp_debug_prop_span(drp_ctx* c, unsigned long long* out, int n) {
    (void)c;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (n > 4096 * 2) n = 4096 * 2;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prop_span), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#endif

int drp_probe_begin(drp_ctx* c, const char* kernel_class) {
    if (!c) return DRP_EINVAL;
    c->probe_cls = -1;
    c->probe_used = 0;
    c->probe_count = false;
    if (!kernel_class || !*kernel_class) return DRP_OK;
    if (strcmp(kernel_class, "prop+work") == 0) { kernel_class = "prop"; c->probe_count = true; }
    for (int i = 0; i < KC_COUNT; ++i)
        if (strcmp(kernel_class, kclass_names[i]) == 0) {
            if (i == KC_PROP && c->probe_count) {
                HIPCHK(c, hipSetDevice(c->device));
                CHK(ensure(c, c->probe_work, PROP_WORK_SHARDS * PROP_WORK_STRIDE * sizeof(unsigned long long)));
                HIPCHK(c, hipMemsetAsync(c->probe_work.p, 0, PROP_WORK_SHARDS * PROP_WORK_STRIDE * sizeof(unsigned long long), c->stream));
            }
            c->probe_cls = i;
            return DRP_OK;
        }
    return fail(c, DRP_EINVAL, "unknown kernel class '%s'", kernel_class);
}

int drp_probe_work(drp_ctx* c, unsigned long long out[8]) {
    if (!c || !out) return fail(c, DRP_EINVAL, "null argument");
    if (c->probe_cls != KC_PROP || !c->probe_count || !c->probe_work.p) return fail(c, DRP_ESTATE, "drp_probe_begin(\"prop+work\") not running");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<unsigned long long> sh((size_t)PROP_WORK_SHARDS * PROP_WORK_STRIDE);
    CHK(d2h(c, sh.data(), c->probe_work.p, sh.size() * sizeof(unsigned long long)));
    CHK(guarded_wait(c, nullptr));
    unsigned long long w[PROP_WORK_COUNT] = {};
    for (int q = 0; q < PROP_WORK_SHARDS; ++q)
        for (int i = 0; i < PROP_WORK_COUNT; ++i) w[i] += sh[(size_t)q * PROP_WORK_STRIDE + i];
    for (int i = 0; i < PROP_WORK_COUNT; ++i) out[i] = w[i];
    // the matrix instructions those units are made of (k_mlp_split.h: the chain of an edge slot, the node layers of a tile)
    out[5] = (unsigned long long)PROP_MFMA_CHAIN * w[PROP_WORK_CHAIN_SLOTS] + (unsigned long long)PROP_MFMA_NODE * w[PROP_WORK_TILES] +
             (unsigned long long)PROP_MFMA_NODE_LAST * w[PROP_WORK_TILES_LAST] + (unsigned long long)PROP_MFMA_ENC * w[PROP_WORK_ENC_TILES];
    // shader-clock cycles and 100 MHz ticks between entry and exit, summed over the workgroups of the counted launches
    out[6] = 0; out[7] = 0;
    for (int q = 0; q < PROP_WORK_SHARDS; ++q) {
        out[6] += sh[(size_t)q * PROP_WORK_STRIDE + PROP_WORK_CLK_CYCLES];
        out[7] += sh[(size_t)q * PROP_WORK_STRIDE + PROP_WORK_CLK_TICKS];
    }
    return DRP_OK;
}

int drp_probe_read(drp_ctx* c, double* total_ms, long* launches) {
    if (!c) return DRP_EINVAL;
    CHK(guarded_wait(c, nullptr));
    double tot = 0.0;
    long n = 0;
    for (size_t i = 0; i + 1 < c->probe_used; i += 2) {
        float ms = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->probe_ev[i], c->probe_ev[i + 1]));
        tot += ms;
        ++n;
    }
    c->probe_used = 0;
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return DRP_OK;
}

// holds the context's stream for `ms` milliseconds (a kernel spinning on the 100 MHz real-time counter): what a
// collective waiting for a dead peer looks like to the host.  tests/test_gpu_errors.py drives the deadline of
// guarded_wait with it.  ms <= 10 000.
__global__ void k_debug_stall(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

int drp_dispatch_reset(drp_ctx* c) {
    if (!c) return DRP_EINVAL;
    memset(c->dv_hit, 0, sizeof(c->dv_hit));
    return DRP_OK;
}

static long dv_join(const unsigned char* hit, bool default_only, char* out, size_t out_len) {
    std::string all;
    char name[96];
    for (int id = 0; id < DV_COUNT; ++id) {
        bool dflt = true;
        dv_name(id, name, sizeof(name), &dflt);
        if (hit ? !hit[id] : (default_only && !dflt)) continue;
        if (!all.empty()) all += ';';
        all += name;
    }
    if (out && out_len) snprintf(out, out_len, "%s", all.c_str());
    return (long)all.size();
}

long drp_last_dispatch(drp_ctx* c, char* out, size_t out_len) {
    if (!c) return DRP_EINVAL;
    return dv_join(c->dv_hit, false, out, out_len);
}

long drp_dispatch_variants(int default_only, char* out, size_t out_len) { return dv_join(nullptr, default_only != 0, out, out_len); }

int drp_range_info(drp_ctx* c, int* shift, double* bound, double* wmax, int* ok) {
    CHK(need(c, true, false, false));
    const SplitRange& r = c->re_range;
    if (shift) *shift = r.shift;
    if (bound) *bound = split_range_bound(r, r.env_attr, r.env_delta, r.env_dens);
    if (wmax) *wmax = (double)r.wmax;
    if (ok) *ok = c->re_ok ? 1 : 0;
    return DRP_OK;
}

int drp_debug_stall(drp_ctx* c, int ms) {
    if (!c || ms < 0 || ms > 10000) return fail(c, DRP_EINVAL, "stall of %d ms outside 0..10000", ms);
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_debug_stall, dim3(1), dim3(1), 0, c->stream, (unsigned long long)ms * 100000ull);
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

long drp_debug_fetch(drp_ctx* c, const char* name, void* out, size_t out_bytes) {
    if (!c || !name || !out) return DRP_EINVAL;
    const size_t bn = (size_t)c->lastB * c->lastN;
    const DevBuf* b = nullptr;
    size_t bytes = 0;
    if (!strcmp(name, "s_delta")) { b = &c->s_delta; bytes = bn * 3 * 4; }
    else if (!strcmp(name, "nbr_idx")) { b = &c->nbr_idx; bytes = bn * DRP_K * 2; }
    else if (!strcmp(name, "nbr_cnt")) { b = &c->nbr_cnt; bytes = bn; }
    else if (!strcmp(name, "effect")) { b = &c->eff; bytes = bn * 64 * 4; }
    else if (!strcmp(name, "c_node")) { b = &c->c_node; bytes = bn * 64 * 4; }
    else if (!strcmp(name, "c_edge")) { b = &c->c_edge; bytes = bn * DRP_K * 64 * 4; }
    else if (!strcmp(name, "proj")) { b = &c->proj; bytes = bn * 128 * 4; }
    else if (!strcmp(name, "agg")) { b = &c->agg; bytes = bn * 64 * 4; }
    else if (!strcmp(name, "stats")) { b = &c->stats; bytes = 8 * sizeof(double); }
    // the blob and its packed copies (tests: the device re-pack after an optimiser step against the host packers)
    else if (!strcmp(name, "w_raw")) { b = &c->w_raw; bytes = (size_t)W_TOTAL * 4; }
    else if (!strcmp(name, "w_valu")) { b = &c->w_valu; bytes = (size_t)V_TOTAL * 4; }
    else if (!strcmp(name, "w_mfma")) { b = &c->w_mfma; bytes = (size_t)M_TOTAL * 4; }
    else if (!strcmp(name, "w_mfma_bwd")) { b = &c->w_mfma_bwd; bytes = (size_t)MB_TOTAL * 4; }
    else if (!strcmp(name, "w_split")) { b = &c->w_split; bytes = (size_t)S_ALLOC * 16; }
    else if (!strcmp(name, "w_split6")) { b = &c->w_split6; bytes = (size_t)S6_TOTAL * 16; }
    else if (!strcmp(name, "w_split6_bwd")) { b = &c->w_split6_bwd; bytes = (size_t)SB6_TOTAL * 16; }
    else if (!strcmp(name, "rev_off")) { b = &c->rev_off; bytes = (size_t)c->lastB * (c->lastN + 1) * 4; }
    else if (!strcmp(name, "rev")) { b = &c->rev; bytes = bn * DRP_K * 4; }
    else return fail(c, DRP_EINVAL, "unknown buffer '%s'", name);
    // a GD session keeps every step's impulses and lists in its tape, not in the step workspace: the last step's
    DevBuf tape{};
    if (c->gd_on && c->gd_H > 0 && bn == (size_t)c->gd_B * c->gd_N) {
        const size_t t = (size_t)c->gd_H - 1;
        if (b == &c->s_delta) { tape.p = ptr<float>(c->tape_sdelta) + t * bn * 3; tape.cap = bytes; b = &tape; }
        else if (b == &c->nbr_idx) { tape.p = ptr<int16_t>(c->tape_idx) + t * bn * DRP_K; tape.cap = bytes; b = &tape; }
        else if (b == &c->nbr_cnt) { tape.p = ptr<uint8_t>(c->tape_cnt) + t * bn; tape.cap = bytes; b = &tape; }
    }
    if (!b->p || bytes == 0 || bytes > b->cap) return fail(c, DRP_ESTATE, "buffer '%s' not populated", name);
    if (out_bytes < bytes) return fail(c, DRP_EINVAL, "buffer '%s' needs %zu bytes", name, bytes);
    if (hipMemcpyAsync(out, b->p, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess)
        return fail(c, DRP_EHIP, "debug fetch failed");
    { const int rc = guarded_wait(c, nullptr); if (rc != DRP_OK) return rc; }
    return (long)bytes;
}
