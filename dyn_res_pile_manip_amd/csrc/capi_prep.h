// capi_prep.h -- a section of the C ABI's translation unit (textually included by drp_capi.hip, in this order: capi_ctx.h,
// capi_pipeline.h, then inside extern "C": capi_core.h, capi_mpc.h, capi_prep.h, capi_gd.h, capi_train.h, capi_comm.h, capi_debug.h).
// Here: particle extraction from the depth image (row f2) and goal pre-processing (row f3).

// ---- particle extraction (row f2) ---------------------------------------------------------------
namespace {
const long long PX_MAX_CELLS = 1ll << 24;
const float PX_FG_DEPTH = (float)(0.599 / 0.8);    // env/flex_env.py:945, compared in float32

int px_nblk(size_t n) { return (int)((n + PX_TILE - 1) / PX_TILE); }

// depth image on the device -> c->px_pcd [n,3] float64 + per-block bounds; *n_out after a sync
int px_stage_pcd(drp_ctx* c, const float* d_depth, const uint8_t* d_mask, int h, int w, float gs, const double cam[4],
                 int* n_out) {
    const size_t npix = (size_t)h * w;
    const int nblk = px_nblk(npix);
    hipStream_t st = c->stream;
    CHK(ensure(c, c->px_blk, (size_t)(2 * nblk + 2) * sizeof(unsigned long long)));
    unsigned long long* cnt = ptr<unsigned long long>(c->px_blk);
    unsigned long long* off = cnt + nblk;
    hipLaunchKernelGGL(k_px_count, dim3(nblk), dim3(PX_BLOCK), 0, st, d_depth, d_mask, gs, PX_FG_DEPTH, npix, cnt);
    hipLaunchKernelGGL(k_px_scan_u64, dim3(1), dim3(1024), 0, st, cnt, nblk, off);
    HIPCHK(c, hipGetLastError());
    unsigned long long total = 0;
    CHK(d2h(c, &total, off + nblk, sizeof(total)));
    CHK(guarded_wait(c, nullptr));
    if (total > 0x7fffffffull) return fail(c, DRP_EINVAL, "too many foreground pixels");
    const int n = (int)total;
    *n_out = n;
    CHK(ensure(c, c->px_pcd, (size_t)(n > 0 ? n : 1) * 3 * sizeof(double)));
    CHK(ensure(c, c->px_bmin, (size_t)nblk * 3 * sizeof(double)));
    CHK(ensure(c, c->px_bmax, (size_t)nblk * 3 * sizeof(double)));
    hipLaunchKernelGGL(k_px_compact, dim3(nblk), dim3(PX_BLOCK), 0, st, d_depth, d_mask, gs, PX_FG_DEPTH, w, npix,
                       cam[0], cam[1], cam[2], cam[3], off, ptr<double>(c->px_pcd), ptr<double>(c->px_bmin),
                       ptr<double>(c->px_bmax));
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

// per-block bounds (c->px_bmin/bmax, nblk blocks) -> voxel grid; cloud d_pcd[n] -> c->px_down[m]
int px_stage_down(drp_ctx* c, const double* d_pcd, int n, int nblk_bounds, double voxel, int* m_out) {
    hipStream_t st = c->stream;
    if (n <= 0) { *m_out = 0; return DRP_OK; }
    CHK(ensure(c, c->px_grid, sizeof(PxGrid)));
    PxGrid* g = ptr<PxGrid>(c->px_grid);
    hipLaunchKernelGGL(k_px_bounds, dim3(1), dim3(64), 0, st, ptr<double>(c->px_bmin), ptr<double>(c->px_bmax),
                       nblk_bounds, n, voxel, g);
    HIPCHK(c, hipGetLastError());
    PxGrid hg;
    CHK(d2h(c, &hg, g, sizeof(hg)));
    CHK(guarded_wait(c, nullptr));
    if (hg.cells <= 0 || hg.cells > PX_MAX_CELLS)
        return fail(c, DRP_EINVAL, "voxel grid %d x %d x %d exceeds %lld cells", hg.dims[0], hg.dims[1], hg.dims[2],
                    PX_MAX_CELLS);
    const long long cells = hg.cells;
    const int cblk = px_nblk((size_t)cells);
    CHK(ensure(c, c->px_keys, (size_t)n * sizeof(int)));
    CHK(ensure(c, c->px_list, (size_t)n * sizeof(int)));
    CHK(ensure(c, c->px_cellcnt, (size_t)cells * sizeof(int)));
    CHK(ensure(c, c->px_cellfill, (size_t)cells * sizeof(int)));
    CHK(ensure(c, c->px_celloff, (size_t)cells * sizeof(unsigned long long)));
    CHK(ensure(c, c->px_blk, (size_t)(2 * cblk + 2) * sizeof(unsigned long long)));
    unsigned long long* bsum = ptr<unsigned long long>(c->px_blk);
    unsigned long long* boff = bsum + cblk;
    HIPCHK(c, hipMemsetAsync(c->px_cellcnt.p, 0, (size_t)cells * sizeof(int), st));
    HIPCHK(c, hipMemsetAsync(c->px_cellfill.p, 0, (size_t)cells * sizeof(int), st));
    const int pblk = (n + 255) / 256;
    hipLaunchKernelGGL(k_px_cell_count, dim3(pblk), dim3(256), 0, st, d_pcd, n, voxel, g, ptr<int>(c->px_keys),
                       ptr<int>(c->px_cellcnt));
    hipLaunchKernelGGL(k_px_cell_blocksum, dim3(cblk), dim3(PX_BLOCK), 0, st, ptr<int>(c->px_cellcnt), cells, bsum);
    hipLaunchKernelGGL(k_px_scan_u64, dim3(1), dim3(1024), 0, st, bsum, cblk, boff);
    hipLaunchKernelGGL(k_px_cell_offsets, dim3(cblk), dim3(PX_BLOCK), 0, st, ptr<int>(c->px_cellcnt), cells, boff,
                       ptr<unsigned long long>(c->px_celloff));
    HIPCHK(c, hipGetLastError());
    unsigned long long total = 0;
    CHK(d2h(c, &total, boff + cblk, sizeof(total)));
    CHK(guarded_wait(c, nullptr));
    const int m = (int)(total >> 32);
    if ((int)(total & 0xffffffffull) != n) return fail(c, DRP_ESTATE, "voxel scan lost points");
    *m_out = m;
    CHK(ensure(c, c->px_down, (size_t)m * 3 * sizeof(double)));
    CHK(ensure(c, c->px_down32, (size_t)m * 3 * sizeof(float)));
    hipLaunchKernelGGL(k_px_cell_fill, dim3(pblk), dim3(256), 0, st, ptr<int>(c->px_keys), n,
                       ptr<unsigned long long>(c->px_celloff), ptr<int>(c->px_cellfill), ptr<int>(c->px_list));
    hipLaunchKernelGGL(k_px_voxel_mean, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, ptr<int>(c->px_cellcnt),
                       ptr<unsigned long long>(c->px_celloff), ptr<int>(c->px_list), d_pcd, cells,
                       ptr<double>(c->px_down), ptr<float>(c->px_down32));
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

// sampler + particle_r (+ recentering radius) on a device cloud (float64 + its float32 copy)
int px_stage_fps(drp_ctx* c, const double* d_pcd, const float* d_pcd32, int m, int npoints, int batch,
                 const int32_t* init_idx, uint64_t seed) {
    hipStream_t st = c->stream;
    const int* d_init = nullptr;
    if (init_idx) {
        for (int b = 0; b < batch; ++b)
            if (init_idx[b] < 0 || init_idx[b] >= m)
                return fail(c, DRP_EINVAL, "init_idx[%d]=%d outside the cloud of %d points", b, init_idx[b], m);
        CHK(h2d(c, c->px_init, init_idx, (size_t)batch * sizeof(int)));
        d_init = ptr<int>(c->px_init);
    }
    CHK(ensure(c, c->px_dist, (size_t)batch * m * sizeof(float)));
    CHK(ensure(c, c->px_chosen, (size_t)batch * npoints * sizeof(int)));
    CHK(ensure(c, c->px_pts, (size_t)batch * npoints * 3 * sizeof(float)));
    CHK(ensure(c, c->px_r, (size_t)batch * sizeof(double)));
    CHK(ensure(c, c->px_rr, (size_t)batch * sizeof(double)));
    hipLaunchKernelGGL(k_px_fps, dim3(batch), dim3(1024), 0, st, d_pcd32, m, npoints, d_init,
                       (unsigned long long)seed, ptr<float>(c->px_dist), ptr<int>(c->px_chosen), ptr<float>(c->px_pts));
    hipLaunchKernelGGL(k_px_radius, dim3(batch), dim3(1024), 0, st, d_pcd, m, ptr<float>(c->px_pts), npoints,
                       ptr<double>(c->px_r), ptr<double>(c->px_rr));
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}

int px_check_cloud(drp_ctx* c, int n, int npoints, int batch) {
    if (npoints <= 0 || batch <= 0) return fail(c, DRP_EINVAL, "bad npoints=%d batch=%d", npoints, batch);
    if (n < npoints) return fail(c, DRP_EINVAL, "cloud of %d points, %d particles asked", n, npoints);
    return DRP_OK;
}
}  // namespace

int drp_depth2fgpcd(drp_ctx* c, const float* depth, const uint8_t* mask, int h, int w, const double cam[4],
                    double* pcd_out, int cap, int* n_out) {
    if (!c || !depth || !cam || !n_out) return fail(c, DRP_EINVAL, "null argument");
    if (h <= 0 || w <= 0) return fail(c, DRP_EINVAL, "bad image size %d x %d", h, w);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t npix = (size_t)h * w;
    CHK(h2d(c, c->px_depth, depth, npix * sizeof(float)));
    if (mask) CHK(h2d(c, c->px_mask, mask, npix));
    int n = 0;
    CHK(px_stage_pcd(c, ptr<float>(c->px_depth), mask ? ptr<uint8_t>(c->px_mask) : nullptr, h, w, 1.0f, cam, &n));
    *n_out = n;
    if (pcd_out) {
        if (cap < n) return fail(c, DRP_EINVAL, "capacity %d < %d foreground points", cap, n);
        if (n > 0) CHK(d2h(c, pcd_out, c->px_pcd.p, (size_t)n * 3 * sizeof(double)));
    }
    return drp_sync(c);
}

int drp_downsample_pcd(drp_ctx* c, const double* pcd, int n, double voxel, double* out, int cap, int* m_out) {
    if (!c || !pcd || !m_out) return fail(c, DRP_EINVAL, "null argument");
    if (n <= 0 || !(voxel > 0.0)) return fail(c, DRP_EINVAL, "bad downsample arguments n=%d voxel=%g", n, voxel);
    HIPCHK(c, hipSetDevice(c->device));
    CHK(h2d(c, c->px_pcd, pcd, (size_t)n * 3 * sizeof(double)));
    const int nblk = px_nblk((size_t)n);
    CHK(ensure(c, c->px_bmin, (size_t)nblk * 3 * sizeof(double)));
    CHK(ensure(c, c->px_bmax, (size_t)nblk * 3 * sizeof(double)));
    hipLaunchKernelGGL(k_px_point_bounds, dim3(nblk), dim3(PX_BLOCK), 0, c->stream, ptr<double>(c->px_pcd), n,
                       ptr<double>(c->px_bmin), ptr<double>(c->px_bmax));
    int m = 0;
    CHK(px_stage_down(c, ptr<double>(c->px_pcd), n, nblk, voxel, &m));
    *m_out = m;
    if (out) {
        if (cap < m) return fail(c, DRP_EINVAL, "capacity %d < %d voxels", cap, m);
        CHK(d2h(c, out, c->px_down.p, (size_t)m * 3 * sizeof(double)));
    }
    return drp_sync(c);
}

int drp_fps_pcd(drp_ctx* c, const double* pcd, int n, int npoints, int batch, const int32_t* init_idx,
                uint64_t seed, float* pts_out, double* r_out) {
    if (!c || !pcd || !pts_out) return fail(c, DRP_EINVAL, "null argument");
    CHK(px_check_cloud(c, n, npoints, batch));
    HIPCHK(c, hipSetDevice(c->device));
    CHK(h2d(c, c->px_down, pcd, (size_t)n * 3 * sizeof(double)));
    CHK(ensure(c, c->px_down32, (size_t)n * 3 * sizeof(float)));
    hipLaunchKernelGGL(k_px_to_f32, dim3((unsigned)(((size_t)n * 3 + 255) / 256)), dim3(256), 0, c->stream,
                       ptr<double>(c->px_down), (size_t)n * 3, ptr<float>(c->px_down32));
    CHK(px_stage_fps(c, ptr<double>(c->px_down), ptr<float>(c->px_down32), n, npoints, batch, init_idx, seed));
    CHK(d2h(c, pts_out, c->px_pts.p, (size_t)batch * npoints * 3 * sizeof(float)));
    if (r_out) CHK(d2h(c, r_out, c->px_r.p, (size_t)batch * sizeof(double)));
    return drp_sync(c);
}

int drp_fps_rad(drp_ctx* c, const double* pcd, int n, double radius, int init_idx, int cap, int32_t* idx_out,
                int* count_out) {
    if (!c || !pcd || !idx_out || !count_out) return fail(c, DRP_EINVAL, "null argument");
    if (n <= 0 || cap <= 0 || init_idx < 0 || init_idx >= n || !(radius >= 0.0))
        return fail(c, DRP_EINVAL, "bad fps_rad arguments n=%d cap=%d init=%d radius=%g", n, cap, init_idx, radius);
    HIPCHK(c, hipSetDevice(c->device));
    CHK(h2d(c, c->px_down, pcd, (size_t)n * 3 * sizeof(double)));
    CHK(ensure(c, c->px_dist, (size_t)n * sizeof(double)));
    CHK(ensure(c, c->px_chosen, (size_t)(cap + 1) * sizeof(int)));
    int* chosen = ptr<int>(c->px_chosen);
    hipLaunchKernelGGL(k_px_fps_rad, dim3(1), dim3(1024), 0, c->stream, ptr<double>(c->px_down), n, radius, init_idx, cap,
                       ptr<double>(c->px_dist), chosen, chosen + cap);
    HIPCHK(c, hipGetLastError());
    CHK(d2h(c, count_out, chosen + cap, sizeof(int)));
    CHK(guarded_wait(c, nullptr));
    CHK(d2h(c, idx_out, chosen, (size_t)*count_out * sizeof(int)));
    return drp_sync(c);
}

int drp_recenter(drp_ctx* c, const double* pcd, int n, const float* sampled, int npoints, int batch, const double* r,
                 float* out) {
    if (!c || !pcd || !sampled || !r || !out) return fail(c, DRP_EINVAL, "null argument");
    if (n <= 0 || npoints <= 0 || batch <= 0) return fail(c, DRP_EINVAL, "bad recenter arguments");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(h2d(c, c->px_down, pcd, (size_t)n * 3 * sizeof(double)));
    CHK(h2d(c, c->px_pts, sampled, (size_t)batch * npoints * 3 * sizeof(float)));
    CHK(h2d(c, c->px_rr, r, (size_t)batch * sizeof(double)));
    CHK(ensure(c, c->px_out, (size_t)batch * npoints * 3 * sizeof(double)));
    float* o32 = ptr<float>(c->px_out);
    hipLaunchKernelGGL(k_px_recenter, dim3((batch * npoints + 3) / 4), dim3(256), 0, c->stream, ptr<double>(c->px_down), n,
                       ptr<float>(c->px_pts), npoints, batch, ptr<double>(c->px_rr), o32, (double*)nullptr);
    HIPCHK(c, hipGetLastError());
    CHK(d2h(c, out, o32, (size_t)batch * npoints * 3 * sizeof(float)));
    return drp_sync(c);
}

int drp_obs2ptcl(drp_ctx* c, const float* depth_raw, int h, int w, float global_scale, const double cam[4],
                 int npoints, int batch, const int32_t* init_idx, uint64_t seed, double* ptcl_out, double* r_out,
                 int* n_fg, int* n_down) {
    if (!c || !depth_raw || !cam || !ptcl_out || !r_out) return fail(c, DRP_EINVAL, "null argument");
    if (h <= 0 || w <= 0 || !(global_scale > 0.0f)) return fail(c, DRP_EINVAL, "bad image %d x %d scale %g", h, w, global_scale);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t npix = (size_t)h * w;
    CHK(h2d(c, c->px_depth, depth_raw, npix * sizeof(float)));
    int n = 0, m = 0;
    CHK(px_stage_pcd(c, ptr<float>(c->px_depth), nullptr, h, w, global_scale, cam, &n));
    if (n_fg) *n_fg = n;
    if (n <= 0) return fail(c, DRP_EINVAL, "no foreground pixel (depth < 0.599/0.8 of the scaled image)");
    CHK(px_stage_down(c, ptr<double>(c->px_pcd), n, px_nblk(npix), 0.01, &m));   // env/flex_env.py:947
    if (n_down) *n_down = m;
    CHK(px_check_cloud(c, m, npoints, batch));
    CHK(px_stage_fps(c, ptr<double>(c->px_down), ptr<float>(c->px_down32), m, npoints, batch, init_idx, seed));
    CHK(ensure(c, c->px_out, (size_t)batch * npoints * 3 * sizeof(double)));
    hipLaunchKernelGGL(k_px_recenter, dim3((batch * npoints + 3) / 4), dim3(256), 0, c->stream, ptr<double>(c->px_down), m,
                       ptr<float>(c->px_pts), npoints, batch, ptr<double>(c->px_rr), (float*)nullptr,
                       ptr<double>(c->px_out));
    HIPCHK(c, hipGetLastError());
    CHK(d2h(c, ptcl_out, c->px_out.p, (size_t)batch * npoints * 3 * sizeof(double)));
    CHK(d2h(c, r_out, c->px_r.p, (size_t)batch * sizeof(double)));
    return drp_sync(c);
}

// ---- goal pre-processing (row f3) ---------------------------------------------------------------
namespace {
// seg (device, [h,w] u8) -> c->gl_dist [h,w] float32
int goal_stage_dt(drp_ctx* c, const uint8_t* d_seg, int h, int w, int mode) {
    const size_t npix = (size_t)h * w;
    hipStream_t st = c->stream;
    CHK(ensure(c, c->gl_tmp, npix * sizeof(int)));
    CHK(ensure(c, c->gl_dist, npix * sizeof(float)));
    if (mode == DRP_DT_CV5) {
        const size_t lds = (size_t)3 * (w + 4) * sizeof(int);
        if (lds > 60000) return fail(c, DRP_EINVAL, "image width %d too large for the chamfer kernel", w);
        c->dv(DV_DT_CV5);
        hipLaunchKernelGGL(k_dt_cv5, dim3(1), dim3(DT_THREADS), lds, st, d_seg, h, w, ptr<int>(c->gl_tmp),
                           ptr<float>(c->gl_dist));
    } else if (mode == DRP_DT_EXACT) {
        if ((size_t)w * sizeof(int) > 60000) return fail(c, DRP_EINVAL, "image width %d too large", w);
        c->dv(DV_DT_EXACT);
        hipLaunchKernelGGL(k_edt_cols, dim3((w + 255) / 256), dim3(256), 0, st, d_seg, h, w, ptr<int>(c->gl_tmp));
        hipLaunchKernelGGL(k_edt_rows, dim3(h), dim3(256), (size_t)w * sizeof(int), st, ptr<int>(c->gl_tmp), h, w,
                           ptr<float>(c->gl_dist));
    } else {
        return fail(c, DRP_EINVAL, "unknown distance transform mode %d", mode);
    }
    HIPCHK(c, hipGetLastError());
    return DRP_OK;
}
}  // namespace

int drp_distance_transform(drp_ctx* c, const uint8_t* src, int h, int w, int mode, float* dist_out) {
    if (!c || !src || !dist_out) return fail(c, DRP_EINVAL, "null argument");
    if (h <= 0 || w <= 0) return fail(c, DRP_EINVAL, "bad image size %d x %d", h, w);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t npix = (size_t)h * w;
    CHK(h2d(c, c->gl_seg, src, npix));
    CHK(goal_stage_dt(c, ptr<uint8_t>(c->gl_seg), h, w, mode));
    CHK(d2h(c, dist_out, c->gl_dist.p, npix * sizeof(float)));
    return drp_sync(c);
}

int drp_set_goal_image(drp_ctx* c, const float* obs_goal, int h, int w, int mode, int max_goal_pts, int fps_init,
                       float* field_out, float* goal_coor_out, int* m_out) {
    if (!c || !obs_goal) return fail(c, DRP_EINVAL, "null argument");
    if (h <= 0 || w <= 0 || max_goal_pts <= 0) return fail(c, DRP_EINVAL, "bad goal image arguments");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t npix = (size_t)h * w;
    const unsigned eb = (unsigned)((npix + 255) / 256);
    CHK(h2d(c, c->gl_goal, obs_goal, npix * sizeof(float)));
    CHK(ensure(c, c->gl_seg, npix));
    hipLaunchKernelGGL(k_goal_seg, dim3(eb), dim3(256), 0, st, ptr<float>(c->gl_goal), npix, ptr<uint8_t>(c->gl_seg));
    // goal pixels first: an image without any is an error before anything is installed
    const int nblk = px_nblk(npix);
    CHK(ensure(c, c->gl_blk, (size_t)(2 * nblk + 2) * sizeof(unsigned long long) + (size_t)(eb + 1) * sizeof(float)));
    unsigned long long* cnt = ptr<unsigned long long>(c->gl_blk);
    unsigned long long* off = cnt + nblk;
    float* bmin = reinterpret_cast<float*>(off + nblk + 2);
    hipLaunchKernelGGL(k_goal_count, dim3(nblk), dim3(PX_BLOCK), 0, st, ptr<uint8_t>(c->gl_seg), npix, cnt);
    hipLaunchKernelGGL(k_px_scan_u64, dim3(1), dim3(1024), 0, st, cnt, nblk, off);
    HIPCHK(c, hipGetLastError());
    unsigned long long total = 0;
    CHK(d2h(c, &total, off + nblk, sizeof(total)));
    CHK(guarded_wait(c, nullptr));
    const int count = (int)total;
    if (count <= 0) return fail(c, DRP_EINVAL, "the goal image has no pixel below 0.5");
    if (count == (int)npix) return fail(c, DRP_EINVAL, "the goal image has no pixel at or above 0.5");
    if (fps_init < 0 || fps_init >= count) return fail(c, DRP_EINVAL, "fps_init=%d outside the %d goal pixels", fps_init, count);
    const int m = max_goal_pts < count ? max_goal_pts : count;
    CHK(ensure(c, c->gl_pix, (size_t)count * 2 * sizeof(float)));
    hipLaunchKernelGGL(k_goal_compact, dim3(nblk), dim3(PX_BLOCK), 0, st, ptr<uint8_t>(c->gl_seg), w, npix, off,
                       ptr<float>(c->gl_pix));
    CHK(ensure(c, c->gl_fps, (size_t)count * sizeof(float) + (size_t)(m + 2) * sizeof(int)));
    float* fdist = ptr<float>(c->gl_fps);
    int* chosen = reinterpret_cast<int*>(fdist + count);
    float* md = reinterpret_cast<float*>(chosen + m);
    c->dv(count <= FPS_WIDE_THREADS * FPS_REG_PT(2) ? DV_FPS_REG : DV_FPS_MEM);
    if (count <= FPS_WIDE_THREADS * FPS_REG_PT(2))
        hipLaunchKernelGGL(k_fps_reg<2>, dim3(1), dim3(FPS_WIDE_THREADS), 0, st, ptr<float>(c->gl_pix), count, m, fps_init, chosen, md);
    else
        hipLaunchKernelGGL(k_fps<2>, dim3(1), dim3(1024), 0, st, ptr<float>(c->gl_pix), count, m, fps_init, fdist, chosen, md);
    CHK(ensure(c, c->goal_coor, (size_t)m * 2 * sizeof(float)));
    hipLaunchKernelGGL(k_goal_gather, dim3((m + 255) / 256), dim3(256), 0, st, ptr<float>(c->gl_pix), chosen, m,
                       ptr<float>(c->goal_coor));
    // the field
    CHK(goal_stage_dt(c, ptr<uint8_t>(c->gl_seg), h, w, mode));
    CHK(ensure(c, c->goal_field, npix * sizeof(float)));
    hipLaunchKernelGGL(k_goal_sub, dim3(eb), dim3(256), 0, st, ptr<float>(c->gl_goal), ptr<float>(c->gl_dist), npix,
                       ptr<float>(c->goal_field), bmin);
    hipLaunchKernelGGL(k_goal_min, dim3(1), dim3(1024), 0, st, bmin, (int)eb, bmin + eb);
    hipLaunchKernelGGL(k_goal_shift, dim3(eb), dim3(256), 0, st, ptr<float>(c->goal_field), npix, bmin + eb);
    HIPCHK(c, hipGetLastError());
    if (field_out) CHK(d2h(c, field_out, c->goal_field.p, npix * sizeof(float)));
    if (goal_coor_out) CHK(d2h(c, goal_coor_out, c->goal_coor.p, (size_t)m * 2 * sizeof(float)));
    CHK(guarded_wait(c, nullptr));
    if (m_out) *m_out = m;
    c->goal_h = h; c->goal_w = w; c->goal_m = m;
    c->have_goal = true;
    return DRP_OK;
}
