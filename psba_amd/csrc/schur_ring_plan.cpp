// schur_ring_plan.cpp -- host-side static schedule of K2's ring route (k_schur_ring).
//
// The route for few cameras (the reference's own data sets and the BAL-sized problems of
// BASELINE.json: 7 ... ~128 cameras).  S_jk -= sum_i Y_ij W_ik^T (reference CL_files/compute_S.cl:24-52,
// compute_Yblks.cl:14-37, compute_ea.cl:26-35) is a scatter when walked point by point and a gather
// when walked block by block.  The LDS-partition route scatters (36 ds_add_f64 per product); this
// route gathers:
//   * the work is cut along two axes: nS stretches of the point sequence x nR ranges of the
//     canonical block order tri(j) + k of the lower block triangle.  One workgroup per (stretch,
//     range); a workgroup's sums stay in registers -- every consumer lane owns ONE 6x6 block (a
//     block has several lanes when it has many products) -- and leave once, into copy `stretch`
//     of the packed triangle.  nS copies instead of one per workgroup: the slab traffic of the
//     LDS-partition route (21 MB written and re-read at 52 cameras) shrinks to nS x 0.4 MB;
//   * operands (the 144-byte W records of the observations a range needs from a point) are
//     streamed into an LDS ring by LDS-DMA (producer waves), Y_a = W_a V*^-1 is formed once per
//     (observation, workgroup) by the producers, and the consumer lanes read both operands of a
//     product from LDS with ds_read_b128;
//   * lanes cannot change blocks, so what a lane does in which step is decided here, once per
//     problem, by simulating the ring: per step every lane takes the next product of its block
//     whose point has arrived, points are loaded in order while the ring has room and are freed
//     in order once all their products are done.  The kernel just replays the lists.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <numeric>
#include <vector>

#include "psba_internal.h"

namespace psba {

namespace {
inline long long tri(long long j) { return j * (j + 1) / 2; }
int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return e && atoi(e) > 0 ? atoi(e) : dflt;
}
}  // namespace

// returns PSBA_OK with out.nWg == 0 when the route does not apply (too many cameras, a track that
// does not fit the ring, ...): the caller then keeps the LDS-partition / owner routes
int build_ring_plan(int nCams, int nPts, int nObs, const int *iidx, const int *jidx, const int *ptr,
                    RingPlanHost &out, bool force) {
  out = RingPlanHost{};
  // Opt-in: measured on MI355X (round 3) the route loses to the LDS-partition
  // route -- its consumers run as planned, but filling LDS with the W records they need (4 x the
  // bytes of W per launch) is bound by what a CU can pull through its vector memory path.
  if (!force && !getenv("PSBA_SCHUR_RING")) return PSBA_OK;
  if (getenv("PSBA_SCHUR_OWNER") || getenv("PSBA_SCHUR_BLOCK_GROUPS")) return PSBA_OK;
  const int max_cams = env_int("PSBA_RING_MAX_CAMS", 128);
  if (nCams > max_cams) return PSBA_OK;
  const long long nBlk = tri(nCams);
  const int NL = RING_LANES, LAT = 2;
  const int budget = env_int("PSBA_RING_SIM_SLOTS", RING_SLOTS);  // (SIM knob: for studying the schedule only)

  // ---- products per block, per point ----
  std::vector<long long> wblk((size_t)nBlk, 0), ppt((size_t)nPts, 0);
  long long total = 0;
  for (int i = 0; i < nPts; i++) {
    const long long k = ptr[i + 1] - ptr[i];
    ppt[(size_t)i] = k * (k + 1) / 2;
    total += ppt[(size_t)i];
    if (k > RING_SLOTS / 4) return PSBA_OK;  // a point must fit the ring beside what is in flight
  }
  for (int a = 0; a < nObs; a++) {
    const long long base = tri(jidx[a]);
    for (int b = ptr[iidx[a]]; b <= a; b++) wblk[(size_t)(base + jidx[b])]++;
  }
  if (total == 0) return PSBA_OK;

  // ---- block ranges: contiguous in canonical order, equal product counts, at most NL / 2 blocks ----
  int nR = 1;
  while (nR < 256 && (nBlk + nR - 1) / nR > NL / 2) nR *= 2;
  if (nCams >= 32 && nR < 16) nR = 16;
  if (const char *e = getenv("PSBA_RING_NR")) nR = atoi(e) > 0 ? atoi(e) : nR;
  if (nR > nBlk) nR = (int)nBlk;
  std::vector<int> rb((size_t)nR + 1, 0);
  {
    // cut by weight, but never more than NL blocks in a range and never an empty range
    long long acc = 0;
    int r = 1;
    for (long long b = 0; b < nBlk && r < nR; b++) {
      acc += wblk[(size_t)b];
      const long long left_blocks = nBlk - (b + 1), left_ranges = nR - r;
      const bool must = (b + 1 - rb[(size_t)r - 1]) >= NL || left_blocks == left_ranges;
      const bool want = (double)acc >= (double)total * r / nR;
      if ((must || want) && b + 1 - rb[(size_t)r - 1] >= 1 && left_blocks >= left_ranges) {
        rb[(size_t)r++] = (int)(b + 1);
      }
    }
    while (r < nR) {  // (degenerate weights) fall back to an even split of what is left
      rb[(size_t)r] = rb[(size_t)r - 1] + 1;
      r++;
    }
    rb[(size_t)nR] = (int)nBlk;
    for (int q = 0; q < nR; q++)
      if (rb[(size_t)q + 1] - rb[(size_t)q] > NL || rb[(size_t)q + 1] <= rb[(size_t)q]) return PSBA_OK;
  }
  std::vector<int> range_of((size_t)nBlk);
  for (int r = 0; r < nR; r++)
    for (int b = rb[(size_t)r]; b < rb[(size_t)r + 1]; b++) range_of[(size_t)b] = r;

  // ---- point stretches, per block range: equal product counts; about 256 workgroups in all, fewer
  // for small problems ----
  int nS = 256 / nR;
  if (nS < 1) nS = 1;
  {
    const long long per_wg = 2048;  // products a workgroup should at least have
    long long cap = total / per_wg / nR;
    if (cap < 1) cap = 1;
    if (cap < nS) nS = (int)cap;
  }
  if (const char *e = getenv("PSBA_RING_NS")) nS = atoi(e) > 0 ? atoi(e) : nS;
  if (nS > nPts) nS = nPts;
  std::vector<long long> rtot((size_t)nR, 0);
  for (long long b = 0; b < nBlk; b++) rtot[(size_t)range_of[(size_t)b]] += wblk[(size_t)b];
  std::vector<int> sp((size_t)nR * (nS + 1), 0);  // range r, stretch s = points [sp[r (nS+1) + s], sp[.. + s + 1])
  {
    std::vector<long long> acc((size_t)nR, 0);
    std::vector<int> nxt((size_t)nR, 1);
    for (int i = 0; i < nPts; i++) {
      for (int a = ptr[i]; a < ptr[i + 1]; a++) {
        const long long base = tri(jidx[a]);
        for (int b = ptr[i]; b <= a; b++) acc[(size_t)range_of[(size_t)(base + jidx[b])]]++;
      }
      for (int r = 0; r < nR; r++)
        while (nxt[(size_t)r] < nS && (double)acc[(size_t)r] >= (double)rtot[(size_t)r] * nxt[(size_t)r] / nS)
          sp[(size_t)r * (nS + 1) + (size_t)nxt[(size_t)r]++] = i + 1;
    }
    for (int r = 0; r < nR; r++) {
      for (int s2 = nxt[(size_t)r]; s2 < nS; s2++) sp[(size_t)r * (nS + 1) + s2] = nPts;
      sp[(size_t)r * (nS + 1) + nS] = nPts;
    }
  }

  out.nR = nR;
  out.nS = nS;
  out.rb = rb;
  out.lat = LAT;

  // row of a block of the canonical order
  auto row_of = [&](long long blk) {
    int j = 0;
    while (tri(j + 1) <= blk) j++;
    return j;
  };

  // ---- per workgroup: lanes, then the simulated schedule ----
  struct Prod { int lb, job, rec; };  // local block; a-side job and partner record (indices into the lists below)
  struct Pt { int i, rec0, nrec, job0, njob, prod0, nprod, loaded; };
  std::vector<Prod> prods;
  std::vector<Pt> pts;
  std::vector<int> precs, pjobs;        // observation of every needed record / a-side job, point by point
  std::vector<int> pjobrec;             // job -> index (into precs) of its own record
  std::vector<int> lane_blk((size_t)NL);
  std::vector<std::vector<int>> queue;  // per local block: product indices in point order
  std::vector<int> qpt;                 // product -> point (index into pts)
  std::vector<int> need;
  std::vector<int> recslot, jobslot, jobpending;
  std::vector<int> slot_pending, slot_free_at, free_y;
  std::vector<char> slot_used;
  std::vector<int> y_release;           // Y slots whose last product was taken in the current step

  out.wgs.clear();
  for (int r = 0; r < nR; r++) {
    const int b0 = rb[(size_t)r], nb = rb[(size_t)r + 1] - b0;
    const int row0 = row_of(b0), row1 = row_of(b0 + nb - 1);
    for (int s = 0; s < nS; s++) {
      // -- the points / records / jobs / products of (r, s) --
      prods.clear();
      pts.clear();
      precs.clear();
      pjobs.clear();
      pjobrec.clear();
      qpt.clear();
      std::vector<long long> w((size_t)nb, 0);
      for (int i = sp[(size_t)r * (nS + 1) + s]; i < sp[(size_t)r * (nS + 1) + s + 1]; i++) {
        const int o0 = ptr[i], o1 = ptr[i + 1];
        if (o1 == o0 || jidx[o1 - 1] < row0 || jidx[o0] > row1) continue;
        Pt p{i, (int)precs.size(), 0, (int)pjobs.size(), 0, (int)prods.size(), 0, -1};
        need.assign((size_t)(o1 - o0), -1);
        // the records this workgroup needs from the point: all partners of its a-side observations
        // lie before them (cameras ascend inside a point), so a PREFIX of the point's records --
        // contiguous in W, which is what lets a page load be one base address (a few records of a
        // row that the range holds only partly are loaded without being used)
        int a_max = -1;
        for (int a = o0; a < o1; a++) {
          const int ja = jidx[a];
          if (ja < row0 || ja > row1) continue;
          const long long base = tri(ja);
          for (int b = o0; b <= a; b++) {
            const long long blk = base + jidx[b];
            if (blk >= b0 && blk < b0 + nb) {
              a_max = a;
              break;
            }
          }
        }
        for (int a = o0; a <= a_max; a++) {
          need[(size_t)(a - o0)] = p.rec0 + p.nrec++;
          precs.push_back(a);
        }
        if (p.nrec == 0) continue;
        for (int a = o0; a < o1; a++) {
          const int ja = jidx[a];
          if (ja < row0 || ja > row1) continue;
          const long long base = tri(ja);
          int job = -1;
          for (int b = o0; b <= a; b++) {
            const long long blk = base + jidx[b];
            if (blk < b0 || blk >= b0 + nb) continue;
            if (job < 0) {
              job = p.job0 + p.njob++;
              pjobs.push_back(a);
              pjobrec.push_back(need[(size_t)(a - o0)]);
            }
            prods.push_back({(int)(blk - b0), job, need[(size_t)(b - o0)]});
            qpt.push_back((int)pts.size());
            w[(size_t)(blk - b0)]++;
            p.nprod++;
          }
        }
        pts.push_back(p);
      }
      // -- lanes per block: every block one, the rest to whoever has the most products per lane --
      std::vector<int> L((size_t)nb, 1);
      for (int left = NL - nb; left > 0; left--) {
        int best = 0;
        for (int b = 1; b < nb; b++)
          if (w[(size_t)b] * L[(size_t)best] > w[(size_t)best] * L[(size_t)b]) best = b;
        if (w[(size_t)best] <= L[(size_t)best]) break;  // nobody has more than one product per lane
        L[(size_t)best]++;
      }
      std::vector<int> lane0((size_t)nb + 1, 0);
      for (int b = 0; b < nb; b++) lane0[(size_t)b + 1] = lane0[(size_t)b] + L[(size_t)b];
      std::fill(lane_blk.begin(), lane_blk.end(), -1);
      for (int b = 0; b < nb; b++)
        for (int l = lane0[(size_t)b]; l < lane0[(size_t)b + 1]; l++) lane_blk[(size_t)l] = b;
      queue.assign((size_t)nb, {});
      for (size_t q = 0; q < prods.size(); q++) queue[(size_t)prods[q].lb].push_back((int)q);
      std::vector<size_t> qhead((size_t)nb, 0);

      // -- the LDS budget of this workgroup: slots for W records and for Y, in the proportion it needs --
      int ny = (int)((double)budget * 1.15 * (double)pjobs.size() / (double)(precs.size() + pjobs.size() + 1));
      if (const char *e = getenv("PSBA_RING_YSLOTS")) ny = atoi(e);
      if (ny < 32) ny = 32;
      if (ny > budget / 2) ny = budget / 2;
      const int nw = budget - ny;  // W slots [0, nw), Y slots behind them
      if ((ny > 0x3FFF || nw > 0xFFFF) && !getenv("PSBA_RING_SIM_SLOTS")) return PSBA_OK;
      slot_pending.assign((size_t)nw, 0);
      slot_free_at.assign((size_t)nw, 0);
      slot_used.assign((size_t)nw, 0);

      RingWg wg{};
      wg.nwslots = nw;
      wg.blk0 = b0;
      wg.nblk = nb;
      wg.row0 = row0;
      wg.nrows = row1 - row0 + 1;
      wg.copy = s;
      wg.step0 = (long long)out.steps.size();
      wg.ent0 = (long long)out.entries.size();
      wg.op0 = (long long)out.ops.size() / 2;
      wg.job0 = (long long)out.jobs.size();
      wg.lane0 = (long long)out.lane_blk.size();
      out.lane_blk.insert(out.lane_blk.end(), lane_blk.begin(), lane_blk.end());
      wg.bl0 = (long long)out.blk_lane0.size();
      out.blk_lane0.insert(out.blk_lane0.end(), lane0.begin(), lane0.end());

      // -- simulate --
      recslot.assign(precs.size(), -1);
      jobslot.assign(pjobs.size(), -1);
      jobpending.assign(pjobs.size(), 0);
      for (const Prod &q : prods) jobpending[(size_t)q.job]++;
      free_y.clear();
      y_release.clear();
      for (int q = ny - 1; q >= 0; q--) free_y.push_back(q);
      int nfree = nw, hint = 0;  // free W slots; where the search for room starts
      size_t next_pt = 0;
      long long done = 0;
      const long long nprod = (long long)prods.size();
      int t = 0;
      for (;; t++) {
        // release the W slots whose records are used up (and whose load / Y preparation is over)
        for (int q = 0; q < nw; q++)
          if (slot_used[(size_t)q] && slot_pending[(size_t)q] == 0 && slot_free_at[(size_t)q] <= t) {
            slot_used[(size_t)q] = 0;
            nfree++;
          }
        for (int ys : y_release) free_y.push_back(ys);
        y_release.clear();
        RingStep st{};
        st.op_begin = (int)(out.ops.size() / 2 - (size_t)wg.op0);
        st.job_begin = (int)(out.jobs.size() - (size_t)wg.job0);
        int ops_now = 0, jobs_now = 0;
        // first fit of `n` consecutive free W slots, searching from `hint` (wrapping once)
        auto find_run = [&](int n) {
          for (int pass = 0; pass < 2; pass++) {
            const int from = pass ? 0 : hint, to = pass ? hint + n - 1 : nw;
            int run = 0;
            for (int q = from; q < to && q < nw; q++) {
              run = slot_used[(size_t)q] ? 0 : run + 1;
              if (run == n) return q - n + 1;
            }
          }
          return -1;
        };
        while (next_pt < pts.size()) {
          Pt &p = pts[next_pt];
          const int pieces = (p.nrec + RING_PAGE - 1) / RING_PAGE;
          if (p.nrec > nfree || p.njob > (int)free_y.size()) break;
          if (ops_now + pieces > RING_MOVERS * RING_MAXOPS || jobs_now + p.njob > RING_PREPPERS * RING_PREP_JOBS) {
            if (ops_now == 0 && jobs_now == 0) return PSBA_OK;  // (a track beyond one step's loads: other routes)
            break;
          }
          // the point's records in runs of at most RING_PAGE (one LDS-DMA instruction each); all or nothing
          std::vector<int> where;
          bool ok = true;
          for (int q = 0; q < p.nrec && ok; q += RING_PAGE) {
            const int n = std::min(RING_PAGE, p.nrec - q);
            const int at = find_run(n);
            if (at < 0) {
              ok = false;
              break;
            }
            for (int k = 0; k < n; k++) slot_used[(size_t)(at + k)] = 1;  // (tentative: undone below if a later run fails)
            where.push_back(at);
          }
          if (!ok) {
            for (size_t w2 = 0; w2 < where.size(); w2++) {
              const int n = std::min(RING_PAGE, p.nrec - (int)w2 * RING_PAGE);
              for (int k = 0; k < n; k++) slot_used[(size_t)(where[w2] + k)] = 0;
            }
            break;  // fragmented: wait for slots to come back
          }
          for (size_t w2 = 0; w2 < where.size(); w2++) {
            const int q0 = (int)w2 * RING_PAGE, n = std::min(RING_PAGE, p.nrec - q0), at = where[w2];
            out.ops.push_back(precs[(size_t)(p.rec0 + q0)]);  // first observation of the run
            out.ops.push_back(at | (n << 16));                 // first slot | records
            for (int k = 0; k < n; k++) {
              recslot[(size_t)(p.rec0 + q0 + k)] = at + k;
              slot_pending[(size_t)(at + k)] = 0;
              slot_free_at[(size_t)(at + k)] = t + LAT;
            }
            hint = at + n;
            ops_now++;
          }
          nfree -= p.nrec;
          for (int q = 0; q < p.njob; q++) {
            const int ys = free_y.back();
            free_y.pop_back();
            jobslot[(size_t)(p.job0 + q)] = ys;
            const int a = pjobs[(size_t)(p.job0 + q)];
            const int ja = jidx[a];
            const long long dblk = tri(ja) + ja;
            RingJob jb{};
            jb.obs = a;
            jb.point = p.i;
            jb.slots = recslot[(size_t)pjobrec[(size_t)(p.job0 + q)]] | (ys << 16);
            jb.earow = (dblk >= b0 && dblk < b0 + nb) ? ja - row0 : -1;
            out.jobs.push_back(jb);
          }
          for (int q = p.prod0; q < p.prod0 + p.nprod; q++)  // every product holds its partner's slot
            slot_pending[(size_t)recslot[(size_t)prods[(size_t)q].rec]]++;
          p.loaded = t;
          jobs_now += p.njob;
          out.loaded_recs += p.nrec;
          next_pt++;
        }
        st.op_end = (int)(out.ops.size() / 2 - (size_t)wg.op0);
        st.job_end = (int)(out.jobs.size() - (size_t)wg.job0);
        // consume: every lane the next product of its block whose point is ready
        const size_t ebase = out.entries.size();
        out.entries.resize(ebase + (size_t)NL, RING_NULL_ENTRY);
        long long consumed_now = 0;
        for (int b = 0; b < nb; b++)
          for (int l = lane0[(size_t)b]; l < lane0[(size_t)b + 1]; l++) {
            if (qhead[(size_t)b] >= queue[(size_t)b].size()) break;
            const int q = queue[(size_t)b][qhead[(size_t)b]];
            const Pt &p = pts[(size_t)qpt[(size_t)q]];
            if (p.loaded < 0 || p.loaded > t - LAT) break;
            qhead[(size_t)b]++;
            const int rs = recslot[(size_t)prods[(size_t)q].rec], ys = jobslot[(size_t)prods[(size_t)q].job];
            out.entries[ebase + (size_t)l] = (unsigned)rs | ((unsigned)ys << 16);
            slot_pending[(size_t)rs]--;
            if (--jobpending[(size_t)prods[(size_t)q].job] == 0) y_release.push_back(ys);
            consumed_now++;
          }
        done += consumed_now;
        out.steps.push_back(st);
        if (done == nprod && next_pt == pts.size()) {
          t++;
          break;
        }
        if (consumed_now == 0 && ops_now == 0 && jobs_now == 0) {
          bool in_flight = false;  // loads whose products cannot start yet
          for (size_t q = 0; q < next_pt && !in_flight; q++)
            if (pts[q].loaded > t - LAT) in_flight = true;
          if (!in_flight) return PSBA_OK;  // (the ring is too small for this problem: other routes)
        }
        if (t > 4000000) return PSBA_OK;
      }
      while (t & 3) {  // the kernel's register sets take turns modulo 4: a multiple of four steps
        RingStep st{};
        st.op_begin = st.op_end = (int)(out.ops.size() / 2 - (size_t)wg.op0);
        st.job_begin = st.job_end = (int)(out.jobs.size() - (size_t)wg.job0);
        out.steps.push_back(st);
        out.entries.resize(out.entries.size() + (size_t)NL, RING_NULL_ENTRY);
        t++;
      }
      wg.nsteps = t;
      out.products += nprod;
      out.slots += (long long)t * NL;
      out.wgs.push_back(wg);
    }
  }
  out.nWg = (int)out.wgs.size();
  // the kernel reads its lists a few steps ahead without bounds checks: slack at the end
  out.entries.resize(out.entries.size() + (size_t)4 * NL, RING_NULL_ENTRY);
  out.ops.resize(out.ops.size() + (size_t)2 * 64 * 2, 0);
  out.jobs.resize(out.jobs.size() + (size_t)RING_PREP_JOBS * RING_PREPPERS + 8, RingJob{0, 0, 0, -1});
  return PSBA_OK;
}

}  // namespace psba
