// capi_mpc.h -- a section of the C ABI's translation unit (textually included by drp_capi.hip, in this order: capi_ctx.h,
// capi_pipeline.h, then inside extern "C": capi_core.h, capi_mpc.h, capi_prep.h, capi_gd.h, capi_train.h, capi_comm.h, capi_debug.h).
// Here: the device-resident sampling planner (drp_mpc_*: sampler, rollout, softmax and elite updates, the one RCCL all-gather) and drp_fps.

// ---- sampling MPC -------------------------------------------------------------------------
int drp_mpc_begin(drp_ctx* c, const drp_mpc_params* p, const float* s0, const float* attr,
                  const float* dens, const double* nominal) {
    CHK(need(c, true, true, true));
    if (!p || !s0 || !attr || !dens || !nominal) return fail(c, DRP_EINVAL, "null argument");
    if (p->n_batch <= 0 || p->n_sample <= 0 || p->n_look_ahead <= 0 || p->n_look_ahead > 64)
        return fail(c, DRP_EINVAL, "bad mpc shape");
    if (p->noise_type < DRP_NOISE_NORMAL || p->noise_type > DRP_NOISE_TOTAL_RAND)
        return fail(c, DRP_EINVAL, "bad noise_type %d", p->noise_type);
    const int nb = p->n_batch, N = p->n_particles, H = p->n_look_ahead, B = p->n_sample * nb;
    CHK(check_bn(c, B, N));
    HIPCHK(c, hipSetDevice(c->device));
    {
        // sampled pushes stay inside the clip box: its two longest diagonals bound every impulse
        const float box[8] = {p->act_lo[0], p->act_lo[1], p->act_hi[2], p->act_hi[3],
                              p->act_hi[0], p->act_hi[1], p->act_lo[2], p->act_lo[3]};
        c->sess_attr_max = max_abs(attr, (size_t)nb * N);
        c->sess_dens_max = max_abs(dens, (size_t)nb);
        CHK(range_check(c, c->sess_attr_max, c->sess_dens_max, push_len_bound(c, box, 2)));
    }
    c->mpc = *p;
    CHK(h2d(c, c->s_in, s0, (size_t)nb * N * 3 * sizeof(float)));
    CHK(h2d(c, c->attr, attr, (size_t)nb * N * sizeof(float)));
    CHK(h2d(c, c->dens, dens, (size_t)nb * sizeof(float)));
    CHK(h2d(c, c->nominal, nominal, (size_t)H * 4 * sizeof(double)));
    CHK(ensure(c, c->actions, (size_t)B * H * 4 * sizeof(float)));
    CHK(ensure(c, c->partials, (size_t)(6 + 4 * H) * sizeof(double)));
    CHK(ensure(c, c->gathered, (size_t)(6 + 4 * H) * sizeof(double) * (size_t)(c->n_ranks > 0 ? c->n_ranks : 1)));
    CHK(ensure(c, c->stats, 8 * sizeof(double)));
    CHK(ensure_step_ws(c, B, N));
    CHK(ensure(c, c->states, (size_t)B * H * N * 3 * sizeof(float)));
    CHK(ensure(c, c->rewards, (size_t)B * H * sizeof(float)));
    CHK(ensure(c, c->scratch, (size_t)B * sizeof(float)));
    CHK(guarded_wait(c, nullptr));
    c->mpc_pending[0] = c->mpc_pending[1] = false;         // a new problem drops what the last one left in flight
    c->mpc_on = true;
    c->gd_on = false;
    c->mpc_cself_tag = 0;           // new attributes / densities / batch size
    return DRP_OK;
}

int drp_mpc_sample(drp_ctx* c, const float* noise, uint64_t iteration) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    HIPCHK(c, hipSetDevice(c->device));
    const drp_mpc_params& p = c->mpc;
    const float* dnoise = nullptr;
    if (noise) {
        CHK(h2d(c, c->noise, noise, (size_t)p.n_sample * p.n_look_ahead * 4 * sizeof(float)));
        dnoise = ptr<float>(c->noise);
    }
    ProbeScope ps(c, KC_MPPI);
    hipLaunchKernelGGL(k_mppi_sample, dim3((4 * p.n_sample + 255) / 256), dim3(256), 0, c->stream,
                       ptr<double>(c->nominal), dnoise, p.n_sample, p.n_batch, p.n_look_ahead, p.sigma,
                       p.beta_filter, make_float4(p.act_lo[0], p.act_lo[1], p.act_lo[2], p.act_lo[3]),
                       make_float4(p.act_hi[0], p.act_hi[1], p.act_hi[2], p.act_hi[3]), p.seed,
                       p.sample_offset, iteration, p.noise_type, ptr<float>(c->actions));
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

int drp_mpc_set_actions(drp_ctx* c, const float* actions) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    HIPCHK(c, hipSetDevice(c->device));
    if (!actions) return fail(c, DRP_EINVAL, "null actions");
    const drp_mpc_params& p = c->mpc;
    CHK(range_check(c, c->sess_attr_max, c->sess_dens_max,
                    push_len_bound(c, actions, (size_t)p.n_sample * p.n_batch * p.n_look_ahead)));
    CHK(h2d(c, c->actions, actions, (size_t)p.n_sample * p.n_batch * p.n_look_ahead * 4 * sizeof(float)));
    return DRP_OK;
}

int drp_mpc_rollout(drp_ctx* c, int reward_all_steps) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    HIPCHK(c, hipSetDevice(c->device));
    const drp_mpc_params& p = c->mpc;
    return run_rollout(c, p.n_batch, p.n_particles, p.n_sample * p.n_batch, p.n_look_ahead,
                       reward_all_steps != 0, reward_all_steps == 0, true);
}

static int launch_partials(drp_ctx* c, double* out) {
    const drp_mpc_params& p = c->mpc;
    const int H = p.n_look_ahead;
    ProbeScope ps(c, KC_MPPI);
    hipLaunchKernelGGL(k_mppi_partials, dim3(4 * H + 1), dim3(256), 0, c->stream,
                       ptr<float>(c->rewards) + (H - 1), H, ptr<float>(c->actions), p.n_sample, p.n_batch,
                       H, p.reward_weight, p.sample_offset, out);
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

int drp_mpc_partials(drp_ctx* c, double* out) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(launch_partials(c, ptr<double>(c->partials)));
    if (out) {
        CHK(d2h(c, out, c->partials.p, (size_t)(6 + 4 * c->mpc.n_look_ahead) * sizeof(double)));
        return drp_sync(c);
    }
    return DRP_OK;
}

static int launch_update(drp_ctx* c, const double* dev_partials, int n_ranks, int rank_stride = 0) {
    const drp_mpc_params& p = c->mpc;
    ProbeScope ps(c, KC_MPPI);
    c->dv(DV_MPPI_SOFTMAX);
    hipLaunchKernelGGL(k_mppi_update, dim3(1), dim3(128), 0, c->stream, dev_partials, n_ranks,
                       rank_stride > 0 ? rank_stride : 6 + 4 * p.n_look_ahead, p.n_look_ahead, (double)p.n_sample * (double)n_ranks, ptr<double>(c->nominal),
                       ptr<double>(c->stats));
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

int drp_mpc_update(drp_ctx* c, const double* partials, int n_ranks, double* nominal_out) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    HIPCHK(c, hipSetDevice(c->device));
    if (!partials || n_ranks <= 0) return fail(c, DRP_EINVAL, "bad partials");
    const size_t rec = (size_t)(6 + 4 * c->mpc.n_look_ahead) * sizeof(double);
    CHK(h2d(c, c->gathered, partials, rec * n_ranks));
    CHK(launch_update(c, ptr<double>(c->gathered), n_ranks));
    if (nominal_out) {
        CHK(d2h(c, nominal_out, c->nominal.p, (size_t)c->mpc.n_look_ahead * 4 * sizeof(double)));
        return drp_sync(c);
    }
    return DRP_OK;
}

int drp_mpc_update_device(drp_ctx* c) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm_failed) return comm_failed_error(c);
    CHK(launch_partials(c, ptr<double>(c->partials)));
    const int rec = 6 + 4 * c->mpc.n_look_ahead;
    if (c->comm && (c->n_ranks > 1 || c->comm_always)) {
        CHK(ensure(c, c->gathered, (size_t)rec * sizeof(double) * c->n_ranks));
        RcclApi* R = rccl_api();
        ncclResult_t r = R->AllGather(c->partials.p, c->gathered.p, rec, ncclDouble, c->comm, c->stream);
        if (r != ncclSuccess) return fail(c, DRP_ECOMM, "ncclAllGather: %s", R->GetErrorString(r));
        return launch_update(c, ptr<double>(c->gathered), c->n_ranks);
    }
    return launch_update(c, ptr<double>(c->partials), 1);
}

// ---- elite (CEM-style) update: nominal = mean of the k best sequences over all ranks
static int elite_check(drp_ctx* c, int k) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    if (k < 1 || k > 1024) return fail(c, DRP_EINVAL, "elite size %d outside 1..1024", k);
    if ((size_t)c->mpc.n_sample * 16 + (size_t)k * 4 > 150 * 1024)
        return fail(c, DRP_EINVAL, "elite update supports up to 9 000 samples per rank");
    return DRP_OK;
}

static int pow2_at_least(int n) { int p = 1; while (p < n) p <<= 1; return p; }

static int launch_elite_local(drp_ctx* c, int k, double* out) {
    const drp_mpc_params& p = c->mpc;
    const int H = p.n_look_ahead;
    // sort path while keys + indices + positions of 2^m >= n_sample entries fit in LDS; k dependent rounds otherwise
    int n2 = pow2_at_least(p.n_sample);
    size_t lds = (size_t)n2 * 20;
    if (lds > 150 * 1024 || k > n2) { n2 = 0; lds = (size_t)p.n_sample * 16 + (size_t)k * 4; }
    ProbeScope ps(c, KC_MPPI);
    c->dv(n2 ? DV_ELITE_SORT : DV_ELITE_ROUNDS);
    hipLaunchKernelGGL(k_elite_local, dim3(1), dim3(256), lds, c->stream, ptr<float>(c->rewards) + (H - 1), H,
                       ptr<float>(c->actions), p.n_sample, p.n_batch, H, k, p.sample_offset, n2, out);
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

static int launch_elite_update(drp_ctx* c, const double* dev_records, int n_ranks, int k, int rank_stride = 0) {
    const int HJ = 4 * c->mpc.n_look_ahead, total = n_ranks * k;
    if (rank_stride <= 0) rank_stride = k * (2 + HJ);
    int n2 = pow2_at_least(total);
    size_t lds = (size_t)n2 * 16 + (size_t)((n2 + 1) / 2) * 8 + (size_t)k * HJ * 8;    // keys, indices, positions, k sequences
    if (lds > 150 * 1024) {
        n2 = 0;
        size_t lds_d = (size_t)total * 2;
        if (lds_d < (size_t)k * HJ) lds_d = (size_t)k * HJ;
        lds = lds_d * 8 + (size_t)k * 4;
        if (lds > 150 * 1024) return fail(c, DRP_EINVAL, "too many elite records (%d ranks x %d, horizon %d)", n_ranks, k, c->mpc.n_look_ahead);
    }
    ProbeScope ps(c, KC_MPPI);
    hipLaunchKernelGGL(k_elite_update, dim3(1), dim3(256), lds, c->stream, dev_records, n_ranks, rank_stride, k, c->mpc.n_look_ahead, n2,
                       ptr<double>(c->nominal), ptr<double>(c->stats) + 6);
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

int drp_mpc_elite(drp_ctx* c, int k, double* out) {
    CHK(elite_check(c, k));
    HIPCHK(c, hipSetDevice(c->device));
    CHK(ensure(c, c->elite, (size_t)k * (2 + 4 * c->mpc.n_look_ahead) * sizeof(double)));
    CHK(launch_elite_local(c, k, ptr<double>(c->elite)));
    if (out) {
        CHK(d2h(c, out, c->elite.p, (size_t)k * (2 + 4 * c->mpc.n_look_ahead) * sizeof(double)));
        return drp_sync(c);
    }
    return DRP_OK;
}

int drp_mpc_update_elite(drp_ctx* c, const double* records, int n_ranks, int k, double* nominal_out) {
    CHK(elite_check(c, k));
    HIPCHK(c, hipSetDevice(c->device));
    if (!records || n_ranks <= 0) return fail(c, DRP_EINVAL, "bad elite records");
    const size_t bytes = (size_t)n_ranks * k * (2 + 4 * c->mpc.n_look_ahead) * sizeof(double);
    CHK(ensure(c, c->elite_all, bytes));
    CHK(h2d(c, c->elite_all, records, bytes));
    CHK(launch_elite_update(c, ptr<double>(c->elite_all), n_ranks, k));
    if (nominal_out) {
        CHK(d2h(c, nominal_out, c->nominal.p, (size_t)c->mpc.n_look_ahead * 4 * sizeof(double)));
        return drp_sync(c);
    }
    return DRP_OK;
}

int drp_mpc_update_elite_device(drp_ctx* c, int k) {
    CHK(elite_check(c, k));
    HIPCHK(c, hipSetDevice(c->device));
    // One message per rank and iteration: [statistics record (6 + 4H) | k elite records (2 + 4H each)],
    // written side by side by the two local kernels, all-gathered with ONE RCCL call, read in place by the
    // two combine kernels (the softmax combine supplies mean / std / max / argmax; its nominal is then replaced
    // by the elite mean).
    if (c->comm_failed) return comm_failed_error(c);
    const int H = c->mpc.n_look_ahead, rec_s = 6 + 4 * H, rec_e = k * (2 + 4 * H), msg = rec_s + rec_e;
    CHK(ensure(c, c->elite, (size_t)msg * sizeof(double)));
    double* mine = ptr<double>(c->elite);
    CHK(launch_partials(c, mine));
    CHK(launch_elite_local(c, k, mine + rec_s));
    const double* all = mine;
    int n_ranks = 1;
    if (c->comm && (c->n_ranks > 1 || c->comm_always)) {
        CHK(ensure(c, c->elite_all, (size_t)msg * sizeof(double) * c->n_ranks));
        RcclApi* R = rccl_api();
        ncclResult_t r = R->AllGather(c->elite.p, c->elite_all.p, msg, ncclDouble, c->comm, c->stream);
        if (r != ncclSuccess) return fail(c, DRP_ECOMM, "ncclAllGather: %s", R->GetErrorString(r));
        all = ptr<double>(c->elite_all);
        n_ranks = c->n_ranks;
    }
    CHK(launch_update(c, all, n_ranks, msg));
    return launch_elite_update(c, all + rec_s, n_ranks, k, msg);
}

int drp_mpc_get(drp_ctx* c, float* actions, float* rewards, float* rewards_all, float* states,
                double* nominal) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    HIPCHK(c, hipSetDevice(c->device));
    const drp_mpc_params& p = c->mpc;
    const int H = p.n_look_ahead, B = p.n_sample * p.n_batch, N = p.n_particles;
    if (actions) CHK(d2h(c, actions, c->actions.p, (size_t)B * H * 4 * sizeof(float)));
    if (rewards)
        HIPCHK(c, hipMemcpy2DAsync(rewards, sizeof(float), ptr<float>(c->rewards) + (H - 1),
                                   H * sizeof(float), sizeof(float), B, hipMemcpyDeviceToHost, c->stream));
    if (rewards_all) CHK(d2h(c, rewards_all, c->rewards.p, (size_t)B * H * sizeof(float)));
    if (states) CHK(d2h(c, states, c->states.p, (size_t)B * H * N * 3 * sizeof(float)));
    if (nominal) CHK(d2h(c, nominal, c->nominal.p, (size_t)H * 4 * sizeof(double)));
    return drp_sync(c);
}

// The planner's loop reads every iteration's pushes and final rewards (planners.py:721-738) but no iteration waits for
// the host: the copies go to pinned memory behind the iteration's kernels (and before the next sampling overwrites the
// pushes), the caller enqueues the next iteration and then waits for this slot's event.
int drp_mpc_fetch_async(drp_ctx* c, int slot) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    if (slot < 0 || slot > 1) return fail(c, DRP_EINVAL, "slot must be 0 or 1");
    if (c->mpc_pending[slot]) return fail(c, DRP_ESTATE, "slot %d holds an iteration nobody has waited for", slot);
    HIPCHK(c, hipSetDevice(c->device));
    const drp_mpc_params& p = c->mpc;
    const int H = p.n_look_ahead, B = p.n_sample * p.n_batch;
    const size_t na = (size_t)B * H * 4, nr = (size_t)B;
    if (c->mpc_pin_floats < na + nr) {
        for (int q = 0; q < 2; ++q) {
            if (c->mpc_pending[q]) return fail(c, DRP_ESTATE, "the batch grew while an iteration was in flight");
            if (c->mpc_pin[q]) HIPCHK(c, hipHostFree(c->mpc_pin[q]));
            c->mpc_pin[q] = nullptr;
            HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->mpc_pin[q]), (na + nr) * sizeof(float), hipHostMallocDefault));
            if (!c->mpc_ev[q]) HIPCHK(c, hipEventCreateWithFlags(&c->mpc_ev[q], hipEventDisableTiming));
        }
        c->mpc_pin_floats = na + nr;
    }
    HIPCHK(c, hipMemcpyAsync(c->mpc_pin[slot], c->actions.p, na * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpy2DAsync(c->mpc_pin[slot] + na, sizeof(float), ptr<float>(c->rewards) + (H - 1), H * sizeof(float),
                               sizeof(float), B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(c->mpc_ev[slot], c->stream));
    c->mpc_pending[slot] = true;
    return DRP_OK;
}

int drp_mpc_wait(drp_ctx* c, int slot, float* actions, float* rewards) {
    if (!c || !c->mpc_on) return fail(c, DRP_ESTATE, "drp_mpc_begin not called");
    if (slot < 0 || slot > 1 || !c->mpc_pending[slot]) return fail(c, DRP_ESTATE, "no iteration in flight in slot %d", slot);
    HIPCHK(c, hipSetDevice(c->device));
    c->mpc_pending[slot] = false;
    CHK(guarded_wait(c, c->mpc_ev[slot]));
    const drp_mpc_params& p = c->mpc;
    const size_t na = (size_t)p.n_sample * p.n_batch * p.n_look_ahead * 4, nr = (size_t)p.n_sample * p.n_batch;
    if (actions) memcpy(actions, c->mpc_pin[slot], na * sizeof(float));
    if (rewards) memcpy(rewards, c->mpc_pin[slot] + na, nr * sizeof(float));
    return DRP_OK;
}

int drp_fps(drp_ctx* c, const float* pts, int n, int dim, int k, int init_idx, int32_t* idx_out, float* max_dist_out) {
    if (!c || !pts || !idx_out) return fail(c, DRP_EINVAL, "null argument");
    if (n <= 0 || k <= 0 || k > n || init_idx < 0 || init_idx >= n || (dim != 2 && dim != 3))
        return fail(c, DRP_EINVAL, "bad fps arguments n=%d dim=%d k=%d init=%d", n, dim, k, init_idx);
    HIPCHK(c, hipSetDevice(c->device));
    CHK(h2d(c, c->scratch, pts, (size_t)n * dim * sizeof(float)));
    CHK(ensure(c, c->g_agg, (size_t)n * sizeof(float) + (size_t)(k + 1) * sizeof(int)));   // dist | chosen | max
    float* dist = ptr<float>(c->g_agg);
    int* chosen = reinterpret_cast<int*>(dist + n);
    CHK(ensure(c, c->stats, 8 * sizeof(double)));
    float* md = reinterpret_cast<float*>(ptr<double>(c->stats) + 7);
    const bool in_regs = n <= FPS_WIDE_THREADS * FPS_REG_PT(dim);
    if (dim == 2) {
        c->dv(in_regs ? DV_FPS_REG : DV_FPS_MEM);
        if (in_regs) hipLaunchKernelGGL(k_fps_reg<2>, dim3(1), dim3(FPS_WIDE_THREADS), 0, c->stream, ptr<float>(c->scratch), n, k, init_idx, chosen, md);
        else hipLaunchKernelGGL(k_fps<2>, dim3(1), dim3(1024), 0, c->stream, ptr<float>(c->scratch), n, k, init_idx, dist, chosen, md);
    } else {
        c->dv(in_regs ? DV_FPS_REG : DV_FPS_MEM);
        if (in_regs) hipLaunchKernelGGL(k_fps_reg<3>, dim3(1), dim3(FPS_WIDE_THREADS), 0, c->stream, ptr<float>(c->scratch), n, k, init_idx, chosen, md);
        else hipLaunchKernelGGL(k_fps<3>, dim3(1), dim3(1024), 0, c->stream, ptr<float>(c->scratch), n, k, init_idx, dist, chosen, md);
    }
    HIPCHK(c, hipGetLastError());
    CHK(d2h(c, idx_out, chosen, (size_t)k * sizeof(int)));
    if (max_dist_out) CHK(d2h(c, max_dist_out, md, sizeof(float)));
    return drp_sync(c);
}
