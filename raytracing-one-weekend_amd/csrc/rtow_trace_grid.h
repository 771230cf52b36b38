// rtow_trace_grid.h — part of the trace kernels (included by rtow_trace_body.h inside namespace rtow::{anonymous};
// see that file for the execution model).  The uniform-grid (3D-DDA) walk.
#pragma once
// --------------------------------------------------------- closest hit: GRID ---
// 3D-DDA over the uniform grid of rtow_grid.h.  Primitives far larger than the rest (the
// ground sphere) are not in the grid; every ray tests that short list first.  Cells are
// visited in order along the ray; a non-empty cell is queued like a BVH leaf and tested in
// the SIMT-dense leaf phase.  The walk ends when the ray leaves the grid (integer cell
// counters, so at most nx+ny+nz steps whatever the floats do) or when the exit distance of
// the current cell is beyond the closest hit so far.
//
// Resumable walk.  A wave's walk lasts as long as its slowest lane — typically one or two rays that
// graze the ground and cross ten cells while the others need one or two.  When the walk has run
// `cap` loop trips and at most `max_open` lanes are still walking, it stops: the queued cells are
// tested, and every lane that is not finished remembers where it stands — the closest hit so far
// (`best`) and the ray parameter at which it entered its current cell (`t_resume` > 0).  The caller
// shades the finished lanes as usual and calls again in its next trip; a resumed lane skips the
// large-primitive list and re-enters the grid at t_resume (a hair earlier: the cell containing that
// point or its predecessor, so no cell is skipped; a cell tested twice changes nothing).  Because of that
// step back, `cap` must be at least 3 (the host clamps it): at most two consecutive cell crossings share one
// ray parameter, so three steps always end in a cell that starts later than the one the lane resumed in.
template <bool LDS, bool ST, int SPEC = 0>
__device__ __forceinline__ Closest closest_hit_grid(const Image<LDS> &im, const DevScene &sc, V3 o,
                                                    V3 d, real time, bool active, uint32_t &nnode,
                                                    uint32_t &nprim, Stamps<ST> &stamps, Closest best,
                                                    float &t_resume, uint32_t cap, uint32_t max_open,
                                                    uint32_t leaf_votes) {
  const bool resumed = t_resume > 0.0f;
  const float t_in = t_resume;
  if (!resumed) {
    best.t = (real)__builtin_huge_val();
    best.prim = -1;
  }
#ifdef RTOW_UNIT_RAYS
  // the walk runs on the unit direction (rtow_trace_bvh.h, RTOW_UNIT_RAYS): its ray parameter is a distance.  The
  // closest hit goes out in the caller's parameter, t = distance / |d|, ONCE: when the segment's walk is complete.  A
  // walk that stops to be resumed keeps `best.t` as the distance it is (the caller does not read it before the walk
  // is complete): converting there and back on every resume would round it differently for every schedule of the
  // wave, and the image must not depend on who traces what next to whom.
  const double a_ref = dot(d, d);
  const double inv_len = fast_rsqrt(a_ref), len = a_ref * inv_len;
  d = d * inv_len;
  const RayForms ray = make_unit_ray_forms(o, d, time, len);
  const float tmin32w = 0.0009f * (float)len;
#else
  const RayForms ray = make_ray_forms(o, d, time);
  const float tmin32w = 0.0009f;
#endif
  ImgOffsets off = {sc.g_off_ids, sc.g_off_sph, sc.g_off_mov, sc.g_off_tri, 0u, 0u, sc.g_off_sph32, sc.g_off_mov32};
  int last_id = -1;
  // header: wave-uniform scalar loads from the global copy of the image
  const RTOW_CONST float *hf = (const RTOW_CONST float *)sc.gblob;
  const RTOW_CONST int32_t *hi = (const RTOW_CONST int32_t *)sc.gblob;
  const float gx = hf[0], gy = hf[1], gz = hf[2];
  const float cx = hf[3], cy = hf[4], cz = hf[5];
  const float icx = hf[6], icy = hf[7], icz = hf[8];
  const int nx = hi[9], ny = hi[10], nz = hi[11];
  const uint32_t n_large = (uint32_t)hi[12], off_large = (uint32_t)hi[13];
  off.fat = (uint32_t)hi[14];
  off.fat_stride = (uint32_t)hi[15];

  // the large primitives, for every ray.  Static spheres are taken four (then two) at a time:
  // all records are loaded and all discriminants computed before any hit branch, so LDS
  // latency and the f64 dependency chains of one test overlap the others.
  if (active && !resumed && n_large != 0u) {
    const uint32_t lf = (off_large - off.ids) >> 2;
    uint32_t k = 0;
    for (; k + 3 < n_large; k += 4) {
      int id[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) id[j] = (int)im.u32(off.ids + 4u * (lf + k + j));
      if (SPEC == 1 || (id[0] < sc.n_sph && id[1] < sc.n_sph && id[2] < sc.n_sph && id[3] < sc.n_sph)) {
        double dd[4], hh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t r = off.sph + 32u * (uint32_t)id[j];
          const double2 p0 = im.d2(r), p1 = im.d2(r + 16u);
          dd[j] = sphere_disc<double>(ray.o64, ray.d64, ray.a64, p0.x, p0.y, p1.x, p1.y, hh[j]);
        }
        nprim += 4u;
#pragma unroll
        for (int j = 0; j < 4; ++j) sphere_resolve<double>(dd[j], hh[j], ray.a64, ray.inv_a64, id[j], ray.tmin, best);
        last_id = id[3];
      } else {
        leaf_test<LDS, false, SPEC>(im, sc, off, lf + k, 4u, ray, best, nprim, last_id);
      }
    }
    for (; k + 1 < n_large; k += 2) {
      const int ia = (int)im.u32(off.ids + 4u * (lf + k)), ib = (int)im.u32(off.ids + 4u * (lf + k + 1));
      if (SPEC == 1 || (ia < sc.n_sph && ib < sc.n_sph)) {
        const uint32_t ra = off.sph + 32u * (uint32_t)ia, rb = off.sph + 32u * (uint32_t)ib;
        const double2 a0 = im.d2(ra), a1 = im.d2(ra + 16u), b0 = im.d2(rb), b1 = im.d2(rb + 16u);
        double ha, hb;
        const double da = sphere_disc<double>(ray.o64, ray.d64, ray.a64, a0.x, a0.y, a1.x, a1.y, ha);
        const double db = sphere_disc<double>(ray.o64, ray.d64, ray.a64, b0.x, b0.y, b1.x, b1.y, hb);
        nprim += 2u;
        sphere_resolve<double>(da, ha, ray.a64, ray.inv_a64, ia, ray.tmin, best);
        sphere_resolve<double>(db, hb, ray.a64, ray.inv_a64, ib, ray.tmin, best);
        last_id = ib;
      } else {
        leaf_test<LDS, false, SPEC>(im, sc, off, lf + k, 2u, ray, best, nprim, last_id);
      }
    }
    if (k < n_large) leaf_test<LDS, false, SPEC>(im, sc, off, lf + k, n_large - k, ray, best, nprim, last_id);
  }
  float tmax32 = round_up_f32(best.t);
  // the walk's set-up — clip against the grid, first cell, the DDA's increments: some eighty binary32 instructions and
  // no memory access — one step below the stage priority (round 5: +0.9 % on the cover scene once the rejection loops
  // were gone and VALU issue utilisation had fallen from 0.96 to 0.89)
  stage_prio<kPrioSetup>();

  // clip the ray to the grid bounds (f32, conservative by the padding of rtow_grid.h)
  const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
  const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
  const float ix = safe_inv(dx), iy = safe_inv(dy), iz = safe_inv(dz);
  const float oix = ox * ix, oiy = oy * iy, oiz = oz * iz;
  const float hx = fmaf((float)nx, cx, gx), hy = fmaf((float)ny, cy, gy), hz = fmaf((float)nz, cz, gz);
  const float ax = fmaf(gx, ix, -oix), bx = fmaf(hx, ix, -oix);
  const float ay = fmaf(gy, iy, -oiy), by = fmaf(hy, iy, -oiy);
  const float az = fmaf(gz, iz, -oiz), bz = fmaf(hz, iz, -oiz);
  const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), fmaxf(tmin32w, t_resume * 0.999999f)));
  const float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax32));
  bool walking = active && t0 <= t1 * 1.00002f;

  // starting cell and DDA state
  const float px = fmaf(t0, dx, ox), py = fmaf(t0, dy, oy), pz = fmaf(t0, dz, oz);
  int c0 = (int)floorf((px - gx) * icx), c1 = (int)floorf((py - gy) * icy), c2 = (int)floorf((pz - gz) * icz);
  c0 = min(max(c0, 0), nx - 1);
  c1 = min(max(c1, 0), ny - 1);
  c2 = min(max(c2, 0), nz - 1);
  const bool fx = dx >= 0.0f, fy = dy >= 0.0f, fz = dz >= 0.0f;
  float tmx = fmaf(fmaf((float)(c0 + (fx ? 1 : 0)), cx, gx), ix, -oix);
  float tmy = fmaf(fmaf((float)(c1 + (fy ? 1 : 0)), cy, gy), iy, -oiy);
  float tmz = fmaf(fmaf((float)(c2 + (fz ? 1 : 0)), cz, gz), iz, -oiz);
  const float tdx = fabsf(cx * ix), tdy = fabsf(cy * iy), tdz = fabsf(cz * iz);
  int remx = fx ? nx - 1 - c0 : c0, remy = fy ? ny - 1 - c1 : c1, remz = fz ? nz - 1 - c2 : c2;
  const int incx = fx ? 1 : -1, incy = fy ? nx : -nx, incz = fz ? nx * ny : -(nx * ny);
  int idx = (c2 * ny + c1) * nx + c0;

  stage_prio<kPrioStage>();
  uint32_t q0 = 0u, q1 = 0u;
  float t_entry = t0;  // ray parameter at which the lane entered its current cell
  uint32_t trips = 0u;  // wave-uniform
  for (;;) {
    if constexpr (ST) {
      stamps.iters += 1;
      const unsigned long long m_step = __ballot(walking && q1 == 0u);
      stamps.step_lanes += (unsigned long long)__popcll(m_step);
      if (m_step != 0ull && (m_step & ~stamps.primary) == 0ull) stamps.iters_cam += 1;
    }
    if (walking && q1 == 0u) {  // (a lane with two cells queued waits for the next leaf phase)
      const uint32_t cw = im.u32(sc.g_off_cells + 4u * (uint32_t)idx);
      ++nnode;
      if (cw != 0u) {
        if (q0 == 0u)
          q0 = cw;
        else
          q1 = cw;
      }
      // leave through the nearest cell wall (x before y before z when equal): one v_min3 and two equality tests
      const float tnext = fminf(fminf(tmx, tmy), tmz);
      const bool sx = tmx == tnext;
      const bool sy = !sx && tmy == tnext;
      const int rem = sx ? remx : (sy ? remy : remz);
      walking = rem > 0 && !(tnext > tmax32);
      t_entry = tnext;
      idx += sx ? incx : (sy ? incy : incz);
      tmx += sx ? tdx : 0.0f;
      tmy += sy ? tdy : 0.0f;
      tmz += (!sx && !sy) ? tdz : 0.0f;
      remx -= sx ? 1 : 0;
      remy -= sy ? 1 : 0;
      remz -= (!sx && !sy) ? 1 : 0;
    }
    const unsigned long long m_walking = __ballot(walking);
    const bool any_walking = m_walking != 0ull;
    ++trips;
    // stop here and resume in the caller's next trip?  (never on the first trips: most lanes finish there)
    // ... never before every resumed lane has got strictly beyond the point it resumed at: progress of a
    // resumed walk is then structural (t_entry takes finitely many f32 values), whatever the rounding does
    // when a ray runs through a cell corner
    const bool suspend = any_walking && trips >= cap && (uint32_t)__popcll(m_walking) <= max_open &&
                         __ballot(walking && resumed && !(t_entry > t_in)) == 0ull;
    // leaf phase: when `leaf_votes` lanes hold a queued cell, when no lane can take a step, or at the end
    const unsigned long long m_pending = __ballot(q0 != 0u);
    if ((m_pending != 0ull && ((uint32_t)__popcll(m_pending) >= leaf_votes || __ballot(walking && q1 == 0u) == 0ull)) ||
        !any_walking || suspend) {
      stamps.mark(RG_WALK, __ballot(active));
      if constexpr (ST) {
        stamps.phases += 1;
        stamps.leaf_lanes += (unsigned long long)__popcll(m_pending);
        if (m_pending != 0ull && (m_pending & ~stamps.primary) == 0ull) stamps.phases_cam += 1;
      }
      stage_prio<kPrioLeaf>();  // a chain of three dependent LDS reads per list entry: first in line for the issue slots
      if (q0 != 0u) leaf_test<LDS, true, SPEC>(im, sc, off, q0 >> 8, q0 & 255u, ray, best, nprim, last_id);
      q0 = q1;
      q1 = 0u;
      if (suspend && __any(q0 != 0u)) {  // both queued cells before stopping
        if (q0 != 0u) leaf_test<LDS, true, SPEC>(im, sc, off, q0 >> 8, q0 & 255u, ray, best, nprim, last_id);
        q0 = 0u;
      }
      tmax32 = round_up_f32(best.t);
      stamps.mark(RG_LEAF, m_pending);
      stage_prio<kPrioStage>();
      if (suspend) {
        // a lane whose next cell starts beyond the hit it has just found is finished after all
        t_resume = (walking && !(t_entry > tmax32)) ? t_entry : 0.0f;
#ifdef RTOW_UNIT_RAYS
        if (!(t_resume > 0.0f)) best.t = best.t * inv_len;  // (the lanes whose walk is complete)
#endif
        return best;
      }
      if (!any_walking && !__any(q0 != 0u)) break;
    }
  }
  t_resume = 0.0f;
#ifdef RTOW_UNIT_RAYS
  best.t = best.t * inv_len;
#endif
  return best;
}

