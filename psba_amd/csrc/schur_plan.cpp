// schur_plan.cpp -- host-side static schedule of K2 (the S / e_a assembly kernel).
//
// The sparsity of a bundle-adjustment problem (which camera sees which point) is fixed for the
// whole LM run, so everything about the scatter  S_jk -= Y_ij W_ik^T  that does not depend on
// values is decided once, at upload time, here (the reference decides the same things per launch
// through blkIdx_buffer look-ups in CL_files/compute_S.cl:13-22 and compute_Y.cl:14-20):
//
//  * S's lower block triangle is split into groups of blocks whose 6x6 accumulators fit in one
//    workgroup's LDS: whole camera rows while 128 groups suffice, else consecutive ranges of the
//    canonical block order tri(j) + k;
//  * one work item per product (a, b), b <= a in the same point; the items of a group, in
//    point-major order, are cut into equally long ranges, one per workgroup, and the groups get
//    workgroups in proportion to their items -- every workgroup has the same amount of work;
//  * where a block lives inside its group's LDS partition is chosen so that the 16 bank pairs
//    of the LDS receive equal traffic, and the items of a workgroup are then dealt into rows of
//    16 lanes whose target blocks fall into 16 different bank pairs: a ds_add_f64
//    wave-instruction over such rows does not serialise (measured, scripts/ubench_lds.hip:
//    8 ticks against 25 for unordered targets).  When no open row can take an item cleanly, a
//    bank pair may be used a second time (never a third): one extra LDS pass for that row
//    instead of a hole -- rows stay short-lived (3 open at a time), which keeps the items of a
//    point, and with them the W rows a wave reads, together.  Remaining holes are null items.
//  * RUNS layout (round 4): real reconstructions list their points so that neighbours share camera sets, and
//    the same block of S then receives products from many consecutive points -- with the rows above those
//    products sit in neighbouring lanes, hit ONE LDS address and are serialised (venice-shaped with runs of 16
//    points per camera set: 78 us against 49 for the uniform draw; 64 points: 105).  When a block's products come
//    in runs (mean run >= 6 over the plan), a workgroup's items are instead dealt as TASKS: the products of
//    one block, in point order, cut into runs of at most RUN_MAX; tasks ordered by their first point and handed
//    to the 512 threads round-robin; a thread sums a task in 36 registers and touches the LDS once per task.
//    Neighbouring threads then work on different blocks of the SAME points, turn after turn.
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

#include "psba_internal.h"

namespace psba {

namespace {
inline long long tri(long long j) { return j * (j + 1) / 2; }
constexpr int ROW = 16;       // lanes that go through the LDS together; bank pairs of a 64-bit access
int DUPS = 1;                 // let a bank pair be used twice in a row once the window is full
int WINDOW = 2;               // open rows while dealing: more rows fill better but spread a point's
                              // products over more waves, i.e. more cache lines per load instruction
                              // (measured on venice-shaped, scripts/k2_window_sweep.sh: strict rows
                              // 80 / 59 / 52 / 50 / 49 us at 1 / 2 / 3 / 4 / 8; with one repeat allowed
                              // 52 / 42 / 42 / 42 / 46 us; end of round 4, A/B on three boxes, assembly + reduce
                              // by HIP events at 2 / 3 / 4 / 6: 48.1-48.4 / 49.0-49.6 / 50.3 / 53.2 -- 2 since then)
}  // namespace

int build_schur_plan(psba_ctx *h, int nCams, int nPts, int nObs, const int *iidx, const int *jidx,
                     const int *ptr, SchurPlanHost &out) {
  const long long total_blocks = tri(nCams);
  if (const char *e = getenv("PSBA_SCHUR_WINDOW")) WINDOW = atoi(e) > 0 ? atoi(e) : 1;
  if (const char *e = getenv("PSBA_SCHUR_DUPS")) DUPS = atoi(e);
  h->packedN = 36 * (size_t)total_blocks;
  h->nGroups = 0;
  if (nCams >= 2048 || getenv("PSBA_SCHUR_OWNER")) return PSBA_OK;  // item fields; PSBA_SCHUR_OWNER forces the owner route
  // ---- row-aligned groups by LDS budget: 37 doubles per block, block count padded to 16 ----
  // 100 KiB rather than all 160: smaller partitions mean less slab traffic (flush + reduce),
  // which on venice-shaped outweighs the loss of locality from more groups (scripts/k2_lds_sweep.sh)
  // (many cameras: the whole LDS rather than giving up on the schedule)
  std::vector<int> lo;
  // (from ~150 cameras on the whole LDS is the better budget at once: fewer groups re-read W less --
  // 200 cameras 62 -> 54 us for assembly + reduce, 257: 69 -> 60)
  for (size_t budget_bytes : {(size_t)(nCams > 150 ? 159 : 100) * 1024, (size_t)159 * 1024}) {
    if (const char *e = getenv("PSBA_SCHUR_LDS_KB")) budget_bytes = (size_t)atoi(e) * 1024;
    const size_t budget_blocks = budget_bytes / sizeof(double) / 37;
    for (int G = 1; G <= MAX_GROUPS && !h->nGroups; G++) {
      lo.assign(1, 0);
      for (int g = 1; g < G; g++) {  // equal-area split of the triangle
        const double target = (double)total_blocks * g / G;
        int j = lo.back();
        while (j < nCams && (double)tri(j) < target) j++;
        if (j <= lo.back()) j = lo.back() + 1;
        if (j > nCams) j = nCams;
        lo.push_back(j);
      }
      lo.push_back(nCams);
      bool ok = true;
      for (int g = 0; g < G && ok; g++) {
        const long long nb = tri(lo[g + 1]) - tri(lo[g]);
        ok = lo[g + 1] > lo[g] && (size_t)((nb + ROW - 1) / ROW * ROW) <= budget_blocks && nb <= 1008;
      }
      if (ok) h->nGroups = G;
    }
    if (h->nGroups) break;
  }
  h->gblk0.clear();
  int split = 1;
  const bool block_ranges = !(h->nGroups && !getenv("PSBA_SCHUR_BLOCK_GROUPS"));
  if (!block_ranges) {
    for (int g = 0; g <= h->nGroups; g++) h->gblk0.push_back((int)tri(lo[g]));
  } else {
    // more cameras than row-aligned groups hold (a camera row alone outgrows a partition from ~550
    // cameras on): consecutive ranges of the canonical block order, whole LDS each.  With at least
    // one workgroup per group the slabs then are one copy of tril(S) whatever the partition size,
    // so the largest partition it is (fewer workgroups, longer item lists each).
    size_t budget_blocks = (size_t)159 * 1024 / sizeof(double) / 37 / ROW * ROW;
    if (const char *e = getenv("PSBA_SCHUR_LDS_KB")) budget_blocks = (size_t)atoi(e) * 1024 / sizeof(double) / 37 / ROW * ROW;
    if (budget_blocks > 1008) budget_blocks = 1008;
    long long G = (total_blocks + (long long)budget_blocks - 1) / (long long)budget_blocks;
    // one workgroup per CU at a time (a partition is most of the LDS): with few rounds, whole rounds
    // of 256 -- 332 groups at 600 cameras ran as one full round and one of 76 (118 us; as 512 smaller
    // groups 99; 800 cameras 118 -> 108; from ~900 groups on rounding up costs more than it balances)
    if (G > 256 && G <= 600 && !getenv("PSBA_SCHUR_NO_ROUNDS")) G = (G + 255) / 256 * 256;
    // a workgroup's stretch of the point sequence must fit the item fields: `split` workgroups (=
    // slabs) per group at least; the slabs (one copy of tril(S) per split) bound what is worth it
    split = 1;
    while ((long long)nObs / split >= (1 << (ITEM_OBS_BITS - 1)) || (long long)nPts / split >= (1 << (ITEM_PT_BITS - 1))) split++;
    if (const char *e = getenv("PSBA_SCHUR_SPLIT")) split = atoi(e) > split ? atoi(e) : split;  // test hook
    double slab_max = 8e9;  // full cfg5 (20 M observations): 10 slabs per group, 5.8 GB, K2 9.6 ms against the owner route's 17.6
    if (const char *e = getenv("PSBA_SCHUR_SLAB_MAX_GB")) slab_max = 1e9 * atof(e);
    if (36 * 8 * (double)total_blocks * split > slab_max) return PSBA_OK;  // beyond what the owner route costs
    h->nGroups = (int)G;
    for (long long g = 0; g <= G; g++) h->gblk0.push_back((int)(total_blocks * g / G));
  }
  const int G = h->nGroups;
  h->gnwg.assign((size_t)G, 1);
  h->gnblk.assign((size_t)G, 0);
  h->gslab.assign((size_t)G, 0);
  std::vector<int> grp_of_blk((size_t)total_blocks);
  for (int g = 0; g < G; g++)
    for (int b = h->gblk0[g]; b < h->gblk0[g + 1]; b++) grp_of_blk[(size_t)b] = g;

  // ---- traffic per block and per group ----
  std::vector<long long> traffic((size_t)total_blocks, 0), gitems((size_t)G, 0);
  for (int a = 0; a < nObs; a++) {
    const long long base = tri(jidx[a]);
    for (int b = ptr[iidx[a]]; b <= a; b++) {
      const size_t blk = (size_t)(base + jidx[b]);
      traffic[blk]++;
      gitems[grp_of_blk[blk]]++;
    }
  }
  const long long total_items = std::accumulate(gitems.begin(), gitems.end(), 0LL);

  // ---- block -> position inside the group's partition: busiest blocks first, each to the
  // bank pair (position mod 16) with the least traffic so far that still has a free position ----
  out.blockpos.assign((size_t)total_blocks, 0);
  for (int g = 0; g < G; g++) {
    const long long b0 = h->gblk0[g], nb = h->gblk0[g + 1] - b0;
    const int per = (int)((nb + ROW - 1) / ROW);
    h->gnblk[g] = per * ROW;
    std::vector<int> order((size_t)nb);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(),
                     [&](int x, int y) { return traffic[(size_t)(b0 + x)] > traffic[(size_t)(b0 + y)]; });
    long long load[ROW] = {0};
    int used[ROW] = {0};
    for (int k : order) {
      int best = -1;
      for (int r = 0; r < ROW; r++)
        if (used[r] < per && (best < 0 || load[r] < load[best])) best = r;
      out.blockpos[(size_t)(b0 + k)] = used[best] * ROW + best;
      used[best]++;
      load[best] += traffic[(size_t)(b0 + k)];
    }
  }

  out.posblock.clear();
  {
    std::vector<size_t> base((size_t)G);
    for (int g = 0; g < G; g++) {
      base[g] = out.posblock.size();
      out.posblock.resize(base[g] + (size_t)h->gnblk[g], -1);
    }
    size_t blk = 0;
    for (int j = 0; j < nCams; j++)
      for (int k = 0; k <= j; k++, blk++) out.posblock[base[grp_of_blk[blk]] + (size_t)out.blockpos[blk]] = (j << 16) | k;
  }

  // ---- workgroups: about one per CU, shared between the groups in proportion to their items ----
  int nWg = 256;
  if (const char *e = getenv("PSBA_SCHUR_NWG")) nWg = atoi(e) > 0 ? atoi(e) : nWg;
  {
    const long long cap = total_items / 512;  // small problems: no point in near-empty workgroups
    if (cap < nWg) nWg = (int)(cap < G ? G : cap);
    if (nWg >= 16) nWg -= nWg % 8;
    if (nWg < G * split) nWg = G * split;  // every group needs `split` of them
  }
  std::vector<int> gnwg((size_t)G, split);
  {
    int left = nWg - G * split;
    std::vector<double> want((size_t)G);
    for (int g = 0; g < G; g++) want[g] = total_items ? (double)gitems[g] * nWg / (double)total_items : 1.0;
    while (left > 0) {  // largest remaining deficit first
      int best = 0;
      for (int g = 1; g < G; g++)
        if (want[g] - gnwg[g] > want[best] - gnwg[best]) best = g;
      gnwg[best]++;
      left--;
    }
  }
  h->nWg = nWg;
  size_t slab_total = 0;
  for (int g = 0; g < G; g++) {
    h->gnwg[g] = gnwg[g];
    h->gslab[g] = slab_total;
    slab_total += (size_t)gnwg[g] * 36 * h->gnblk[g];
  }
  out.slab_doubles = slab_total;

  // ---- items of each group in point-major order, cut into gnwg[g] equal ranges ----
  struct Raw { int a, i, boff, pos; };
  std::vector<std::vector<Raw>> raw((size_t)G);
  for (int g = 0; g < G; g++) raw[g].reserve((size_t)gitems[g]);
  for (int a = 0; a < nObs; a++) {
    const int ja = jidx[a], i = iidx[a];
    if (a - ptr[i] >= (1 << ITEM_BOFF_BITS)) {  // (psba_upload_problem refuses such tracks anyway)
      h->nGroups = 0;
      return PSBA_OK;
    }
    for (int b = ptr[i]; b <= a; b++) {
      const size_t blk = (size_t)(tri(ja) + jidx[b]);
      raw[grp_of_blk[blk]].push_back({a, i, a - b, out.blockpos[blk]});
    }
  }
  // runs of one block over consecutive products of a group's point-major list: how clustered are the tracks?
  long long n_tasks_probe = 0;
  for (int g = 0; g < G; g++) {
    std::vector<int> last((size_t)h->gnblk[g], -2), len((size_t)h->gnblk[g], 0);
    int prev_pt = -1, pt_ord = -1;
    for (const Raw &it : raw[g]) {
      if (it.i != prev_pt) {
        prev_pt = it.i;
        pt_ord++;
      }
      // a run continues when the block's previous product came from the previous point that touches this group
      if (last[(size_t)it.pos] != pt_ord - 1 && last[(size_t)it.pos] != pt_ord) n_tasks_probe++, len[(size_t)it.pos] = 0;
      if (++len[(size_t)it.pos] > RUN_MAX) n_tasks_probe++, len[(size_t)it.pos] = 1;
      last[(size_t)it.pos] = pt_ord;
    }
  }
  // (venice-shaped, assembly + reduce, rows / runs layout: runs of 4 points 59 / 68 us, 16: 78 / 59, 64: 105 / 57; the
  // uniform draw 49 / 74 -- the runs kernel runs two waves per SIMD, so it only pays once runs are long)
  out.runs = n_tasks_probe > 0 && (double)total_items / (double)n_tasks_probe >= 6.0 && !block_ranges;
  if (const char *e = getenv("PSBA_SCHUR_RUNS")) out.runs = atoi(e) != 0 && !block_ranges;
  out.tasks = 0;
  out.pair_items = 0;
  // pair items (one observation, two partners): built and measured in round 4 -- 41.5 against
  // 38.5 us on venice-shaped -- so only the experiments build deals them, on request
  bool use_pairs = false;
#ifdef PSBA_BUILD_EXPERIMENTS
  if (const char *e = getenv("PSBA_SCHUR_PAIRS")) use_pairs = atoi(e) != 0 && !block_ranges && !out.runs;
#endif
  struct WgTmp { SchurWg w; double where; };
  std::vector<WgTmp> wgs;
  out.items.clear();
  for (int g = 0; g < G; g++) {
    const size_t n = raw[g].size();
    for (int k = 0; k < gnwg[g]; k++) {
      const size_t r0 = n * k / gnwg[g], r1 = n * (k + 1) / gnwg[g];
      SchurWg w{};
      w.group = g;
      w.nblk = h->gnblk[g];
      w.slab_off = h->gslab[g] + (size_t)k * 36 * h->gnblk[g];
      w.obs0 = r1 > r0 ? raw[g][r0].a : 0;
      w.pt0 = r1 > r0 ? raw[g][r0].i : 0;
      if (r1 > r0 && (raw[g][r1 - 1].a - w.obs0 >= (1 << ITEM_OBS_BITS) || raw[g][r1 - 1].i - w.pt0 >= (1 << ITEM_PT_BITS))) {
        h->nGroups = 0;  // the item encoding does not hold this range: owner route instead
        return PSBA_OK;
      }
      const size_t base = out.items.size();
      if (out.runs) {
        // tasks: per block position the range's products in order, cut at RUN_MAX; ordered by first product
        struct Task { size_t first; std::vector<size_t> idx; };
        std::vector<Task> tasks;
        // (a small range: shorter runs, so that every thread gets a few of them)
        const size_t run_cap = std::min<size_t>((size_t)RUN_MAX, std::max<size_t>(2, (r1 - r0) / (2 * RUN_THREADS)));
        {
          std::vector<int> open((size_t)w.nblk, -1);
          for (size_t t = r0; t < r1; t++) {
            const int pos = raw[g][t].pos;
            if (open[(size_t)pos] < 0 || tasks[(size_t)open[(size_t)pos]].idx.size() >= run_cap) {
              open[(size_t)pos] = (int)tasks.size();
              tasks.push_back({t, {}});
            }
            tasks[(size_t)open[(size_t)pos]].idx.push_back(t);
          }
        }
        out.tasks += (long long)tasks.size();
        // (already ordered by first product: a task is created when its first product is met)
        // round-robin over the threads in task order: the tasks that begin at the same points (one per block of a
        // run of points with one camera set) land in neighbouring threads and read the same W rows turn by turn
        // (... but a thread that is ahead of the others is passed over: the lists stay within a run of each other)
        std::vector<std::vector<unsigned long long>> list((size_t)RUN_THREADS);
        size_t thr = RUN_THREADS - 1, level = 0, scanned = 0;
        for (size_t q = 0; q < tasks.size(); q++) {
          do {
            thr = (thr + 1) % RUN_THREADS;
            if (++scanned > (size_t)RUN_THREADS) {
              level += 2;
              scanned = 0;
            }
          } while (list[thr].size() > level);
          scanned = 0;
          for (size_t t : tasks[q].idx) {
            const Raw &it = raw[g][t];
            list[thr].push_back((unsigned long long)(it.a - w.obs0) | ((unsigned long long)(it.i - w.pt0) << ITEM_OBS_BITS) |
                                ((unsigned long long)it.boff << (ITEM_OBS_BITS + ITEM_PT_BITS)) |
                                ((unsigned long long)it.pos << (ITEM_OBS_BITS + ITEM_PT_BITS + ITEM_BOFF_BITS)));
          }
        }
        size_t T = 0;
        for (const auto &l : list) T = l.size() > T ? l.size() : T;
        out.items.resize(base + T * RUN_THREADS, SCHUR_NULL_ITEM);
        for (int thr = 0; thr < RUN_THREADS; thr++)
          for (size_t t = 0; t < list[(size_t)thr].size(); t++) out.items[base + t * RUN_THREADS + thr] = list[(size_t)thr][t];
        w.item0 = w.itemD = (long long)base;
        w.item1 = (long long)out.items.size();
        wgs.push_back({w, n ? (double)(r0 + r1) / (2.0 * (double)n) : 0.0});
        continue;
      }
      // deal the range into rows of 16 with distinct bank pairs (first fit over a window of open rows).  A slot
      // has one block position (one product) or two (a pair item: two products, added in two phases -- the 16
      // lanes must hit 16 different bank pairs in each phase)
      struct Slot { unsigned long long enc; int pos[2]; int np; };
      // block ranges (many cameras): a group holds at most one diagonal block, which alone receives
      // as many items as all its other blocks together (every observation's self-product).  Lanes on
      // one address are serialised by the LDS whichever rows they sit in, so an item may always
      // join a row that already has its very position: without this such lists are rows of two
      // items and fourteen holes (fill 0.54 at cfg5)
      const bool same_pos_ok = block_ranges;
      auto deal = [&](const std::vector<Slot> &slots) {
        const size_t base_rows = out.items.size();
        size_t closed = 0;  // rows [0, closed) are final
        std::vector<uint16_t> mask[2], mask2[2];  // per phase: bank pairs used once / twice in a row
        std::vector<int> fill;
        std::vector<std::vector<int>> rowpos;
        auto row_ptr = [&](size_t r) { return out.items.data() + base_rows + r * ROW; };
        for (const Slot &sl : slots) {
          uint16_t bit[2] = {(uint16_t)(1u << (sl.pos[0] % ROW)), (uint16_t)(sl.np > 1 ? 1u << (sl.pos[1] % ROW) : 0)};
          auto has_pos = [&](size_t row) {
            if (!same_pos_ok || sl.np > 1) return false;
            for (int q : rowpos[row])
              if (q == sl.pos[0]) return true;
            return false;
          };
          auto clean = [&](size_t row) {
            for (int ph = 0; ph < sl.np; ph++)
              if ((mask[ph][row] & bit[ph]) && !has_pos(row)) return false;
            return true;
          };
          auto once_more = [&](size_t row) {  // every phase either clean or a first repeat of its bank pair
            for (int ph = 0; ph < sl.np; ph++)
              if ((mask[ph][row] & bit[ph]) && (mask2[ph][row] & bit[ph])) return false;
            return true;
          };
          size_t r = closed;
          const size_t nrows = fill.size();
          while (r < nrows && (fill[r] == ROW || !clean(r))) r++;
          if (r == nrows && DUPS && nrows - closed >= (size_t)WINDOW) {
            // no clean slot and the window is full: rather than opening a row (and closing the
            // oldest one with holes), let a bank pair be used twice -- one extra LDS pass for
            // that row of 16 lanes instead of idle lanes
            r = closed;
            while (r < nrows && (fill[r] == ROW || !once_more(r))) r++;
          }
          if (r == nrows) {
            for (int ph = 0; ph < 2; ph++) {
              mask[ph].push_back(0);
              mask2[ph].push_back(0);
            }
            fill.push_back(0);
            rowpos.emplace_back();
            out.items.resize(base_rows + fill.size() * ROW, SCHUR_NULL_ITEM);
          }
          row_ptr(r)[fill[r]++] = sl.enc;
          for (int ph = 0; ph < sl.np; ph++) {
            if (mask[ph][r] & bit[ph]) mask2[ph][r] |= bit[ph];
            mask[ph][r] |= bit[ph];
          }
          if (same_pos_ok) rowpos[r].push_back(sl.pos[0]);
          while (closed < fill.size() && (fill[closed] == ROW || fill.size() - closed > (size_t)WINDOW)) closed++;
        }
      };
      // pair items (round 4): consecutive products of one observation a (partners b, b + 1) share W_a, V*^-1, Y_a
      // and the e_a terms.  Row-aligned groups keep all partners of an observation in its camera's group; the
      // range's extent must fit the pair fields.
      const bool pairs = use_pairs && r1 > r0 && raw[g][r1 - 1].a - w.obs0 < (1 << PAIR_OBS_BITS) &&
                         raw[g][r1 - 1].i - w.pt0 < (1 << PAIR_PT_BITS);
      std::vector<Slot> dbl, sgl;
      size_t run_end = r0;  // end of the products of it.a inside the range
      bool odd_done = false;
      for (size_t t = r0; t < r1; t++) {
        const Raw &it = raw[g][t];
        if (t == run_end) {
          while (run_end < r1 && raw[g][run_end].a == it.a) run_end++;
          odd_done = false;
        }
        // an odd number of products: the FIRST one goes alone (the last one is the self-product, and the diagonal
        // blocks of a group are too few positions to fill rows of 16 bank pairs with)
        const bool alone = ((run_end - t) & 1) && !odd_done;
        if (alone) odd_done = true;
        if (pairs && !alone && t + 1 < run_end) {
          const Raw &nx = raw[g][t + 1];  // (the list runs b upwards: nx.boff == it.boff - 1)
          dbl.push_back({(unsigned long long)(it.a - w.obs0) | ((unsigned long long)(it.i - w.pt0) << PAIR_OBS_BITS) |
                             ((unsigned long long)it.boff << (PAIR_OBS_BITS + PAIR_PT_BITS)) |
                             ((unsigned long long)it.pos << (PAIR_OBS_BITS + PAIR_PT_BITS + ITEM_BOFF_BITS)) |
                             ((unsigned long long)nx.pos << (PAIR_OBS_BITS + PAIR_PT_BITS + ITEM_BOFF_BITS + ITEM_POS_BITS)),
                         {it.pos, nx.pos}, 2});
          t++;
          continue;
        }
        sgl.push_back({(unsigned long long)(it.a - w.obs0) | ((unsigned long long)(it.i - w.pt0) << ITEM_OBS_BITS) |
                           ((unsigned long long)it.boff << (ITEM_OBS_BITS + ITEM_PT_BITS)) |
                           ((unsigned long long)it.pos << (ITEM_OBS_BITS + ITEM_PT_BITS + ITEM_BOFF_BITS)),
                       {it.pos, 0}, 1});
      }
      deal(dbl);
      w.itemD = (long long)out.items.size();
      deal(sgl);
      out.pair_items += (long long)dbl.size();
      w.item0 = (long long)base;
      w.item1 = (long long)out.items.size();
      wgs.push_back({w, n ? (double)(r0 + r1) / (2.0 * (double)n) : 0.0});
    }
  }
  // workgroups ordered by where in the point sequence they work: neighbours (which re-read the
  // same W rows for different groups) land on the same XCD, see k_schur_lds
  std::stable_sort(wgs.begin(), wgs.end(), [](const WgTmp &x, const WgTmp &y) { return x.where < y.where; });
  out.wgs.clear();
  for (auto &t : wgs) out.wgs.push_back(t.w);
  out.real_items = total_items;
  return PSBA_OK;
}

// ---- owner route (many cameras) -------------------------------------------------------------
// Products (a, b), b <= a in the same point, are bucketed by target block (counting sort, so a
// block's list stays point-major); lists longer than LMAX are cut into segments (units); units
// are sorted by length (longest first) and dealt 64 to a wave; a wave's products go into
// ELL rows [t][lane].  LMAX aims at >= 256 k units so that every SIMD has several waves, but not
// below 16 products per unit (shorter units would only multiply the atomic adds of the combine).
int build_owner_plan(int nCams, int nObs, const int *iidx, const int *jidx, const int *ptr, OwnerPlanHost &out,
                     const unsigned char *pattern) {
  const long long nBlk = tri(nCams);
  std::vector<long long> cnt((size_t)nBlk + 1, 0);
  long long total = 0;
  for (int a = 0; a < nObs; a++) {
    const long long base = tri(jidx[a]);
    for (int b = ptr[iidx[a]]; b <= a; b++) cnt[(size_t)(base + jidx[b]) + 1]++;
    total += a - ptr[iidx[a]] + 1;
  }
  for (long long t = 0; t < nBlk; t++) cnt[(size_t)t + 1] += cnt[(size_t)t];
  std::vector<int2> sorted((size_t)total);
  {
    std::vector<long long> at(cnt.begin(), cnt.end() - 1);
    for (int a = 0; a < nObs; a++) {
      const long long base = tri(jidx[a]);
      for (int b = ptr[iidx[a]]; b <= a; b++) sorted[(size_t)at[(size_t)(base + jidx[b])]++] = make_int2(a, b);
    }
  }
  long long LMAX = (total + 262143) / 262144;
  if (LMAX < 16) LMAX = 16;
  if (const char *e = getenv("PSBA_OWNER_LMAX")) LMAX = atoll(e) > 0 ? atoll(e) : LMAX;
  // the blocks that exist: every diagonal block, and the off-diagonal ones with at least one product --
  // here, or (sharded points: `pattern`, one byte per block of the lower triangle) on any rank, so that
  // the block lists of all ranks are the same list
  out.blocks.clear();
  out.diag_slot.assign((size_t)nCams, 0);
  std::vector<int> slot_of((size_t)nBlk, -1);
  for (int j = 0; j < nCams; j++)
    for (int k = 0; k <= j; k++) {
      const long long blk = tri(j) + k;
      if (k == j || cnt[(size_t)blk + 1] > cnt[(size_t)blk] || (pattern && pattern[(size_t)blk])) {
        slot_of[(size_t)blk] = (int)out.blocks.size();
        if (k == j) out.diag_slot[(size_t)j] = (int)out.blocks.size();
        out.blocks.push_back(make_int2(j, k));
      }
    }
  struct Unit { long long first; int len, j, k, multi; };
  std::vector<Unit> units;
  units.reserve((size_t)(total / LMAX + nBlk));
  for (int j = 0; j < nCams; j++)
    for (int k = 0; k <= j; k++) {
      const long long blk = tri(j) + k, n = cnt[(size_t)blk + 1] - cnt[(size_t)blk];
      if (n == 0) continue;
      const long long pieces = (n + LMAX - 1) / LMAX;
      for (long long q = 0; q < pieces; q++) {
        const long long f = n * q / pieces, l = n * (q + 1) / pieces;
        units.push_back({cnt[(size_t)blk] + f, (int)(l - f), j, k, pieces > 1 ? 1 : 0});
      }
    }
  std::stable_sort(units.begin(), units.end(), [](const Unit &x, const Unit &y) { return x.len > y.len; });
  const size_t nW = (units.size() + 63) / 64;
  out.waves.resize(nW);
  out.units.assign(nW * 64, OwnerUnit{0, 0, 0, 0});
  long long rows = 0;
  for (size_t w = 0; w < nW; w++) {
    const int len = units[w * 64].len;  // sorted: the first unit of a wave is its longest
    out.waves[w] = {rows, len, 0};
    rows += len;
  }
  out.prod.assign((size_t)rows * 64, make_int2(-1, -1));
  for (size_t u = 0; u < units.size(); u++) {
    const size_t w = u / 64, lane = u % 64;
    out.units[u] = {units[u].j, units[u].k, units[u].multi, slot_of[(size_t)(tri(units[u].j) + units[u].k)]};
    int2 *dst = out.prod.data() + (size_t)out.waves[w].row0 * 64 + lane;
    for (int t = 0; t < units[u].len; t++) dst[(size_t)t * 64] = sorted[(size_t)(units[u].first + t)];
  }
  for (size_t u = units.size(); u < nW * 64; u++) out.units[u] = {0, 0, 1, -1};  // idle lanes (slot = -1)
  out.products = total;
  return PSBA_OK;
}

// one byte per block tri(j) + k of the lower block triangle: 1 where cameras j and k see a common point
int sparse_pattern(int nCams, int nObs, const int *iidx, const int *jidx, const int *ptr, unsigned char *flags) {
  const long long nBlk = tri(nCams);
  for (long long t = 0; t < nBlk; t++) flags[t] = 0;
  for (int a = 0; a < nObs; a++) {
    const long long base = tri(jidx[a]);
    for (int b = ptr[iidx[a]]; b <= a; b++) flags[base + jidx[b]] = 1;
  }
  return PSBA_OK;
}

}  // namespace psba
