// sba_io.cpp -- reader, writer and BAL converter for the sba text format (host only).
// Restates readInitialSBAEstimate and its helpers (reference PSBA/readparams.cpp:169-232,
// 247-290,332-423,444-519) together with the quat2vec input filter (PSBA/misc.cpp:21-49) and
// the parameter split the driver performs (PSBA/main.cpp:131-149; fixed-K variant
// PSBA/main_bak.cpp:32,65-71).  Format: SURVEY.md Appendix C.
//   cams file: one camera per line, '#' comments; 7 columns (q0..q3 tx ty tz), 12 columns
//              (fu u0 v0 ar s + those 7) or 17 columns (+5 distortion terms, ignored: the
//              reference never optimises or applies them, main.cpp:73,102-103).
//   pts file : X Y Z nframes { frame x y [cov] } ...; covariance (4 or 3 values) is
//              detected from the first line and skipped (reference reads, then ignores it,
//              main.cpp:112).
// Difference kept on purpose: frames of a point are sorted by camera id.  The reference
// stores projections in file order but indexes them in camera order (readparams.cpp:364-373
// vs misc.cpp:189-216) and so silently needs ascending ids; every bundled file has them.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/psba_hip.h"

namespace {

bool read_data_line(FILE *fp, std::string &line) {
  line.clear();
  char buf[4096];
  while (true) {
    bool got = false;
    std::string cur;
    while (fgets(buf, sizeof buf, fp)) {
      got = true;
      cur += buf;
      if (!cur.empty() && cur.back() == '\n') break;
    }
    if (!got) return false;
    size_t k = cur.find_first_not_of(" \t\r\n");
    if (k == std::string::npos || cur[k] == '#') continue;
    line.swap(cur);
    return true;
  }
}

bool parse_doubles(const std::string &s, std::vector<double> &out) {
  out.clear();
  const char *p = s.c_str();
  char *end = nullptr;
  while (true) {
    double v = strtod(p, &end);
    if (end == p) break;
    out.push_back(v);
    p = end;
  }
  while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') p++;
  return *p == '\0';
}

template <typename T>
T *dup(const std::vector<T> &v) {
  T *p = (T *)malloc(sizeof(T) * (v.empty() ? 1 : v.size()));
  if (p && !v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
  return p;
}

}  // namespace

extern "C" {

void psba_free_problem(psba_problem *p) {
  if (!p) return;
  free(p->Kparas);
  free(p->impts);
  free(p->initrot);
  free(p->camsEx);
  free(p->pts3D);
  free(p->iidx);
  free(p->jidx);
  memset(p, 0, sizeof *p);
}

int psba_read_problem(const char *cams_file, const char *pts_file, const double *fixedK,
                      psba_problem *out) {
  if (!cams_file || !pts_file || !out) return PSBA_E_INVALID;
  memset(out, 0, sizeof *out);
  FILE *fc = fopen(cams_file, "r");
  if (!fc) return PSBA_E_IO;
  FILE *fp = fopen(pts_file, "r");
  if (!fp) {
    fclose(fc);
    return PSBA_E_IO;
  }
  std::vector<double> K, rot, cams, pts, impts, vals;
  std::vector<int> iidx, jidx;
  std::string line;
  int rc = PSBA_OK, ncol = -1, nC = 0;
  while (rc == PSBA_OK && read_data_line(fc, line)) {
    if (!parse_doubles(line, vals)) { rc = PSBA_E_IO; break; }
    if (ncol < 0) ncol = (int)vals.size();
    if ((int)vals.size() != ncol || !(ncol == 7 || ncol == 12 || ncol == 17)) { rc = PSBA_E_IO; break; }
    const double *q = vals.data() + (ncol - 7);
    if (ncol == 7) {
      if (!fixedK) { rc = PSBA_E_INVALID; break; }
      K.insert(K.end(), fixedK, fixedK + 5);
    } else {
      K.insert(K.end(), vals.begin(), vals.begin() + 5);
    }
    // quat2vec (misc.cpp:38-43): normalise, force a non-negative scalar part
    const double mag = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double sg = (q[0] >= 0.0) ? 1.0 : -1.0;
    const double f = sg / mag;
    const double v0 = q[1] * f, v1 = q[2] * f, v2 = q[3] * f;
    // initrot (readparams.cpp:223-226)
    rot.push_back(std::sqrt(1.0 - v0 * v0 - v1 * v1 - v2 * v2));
    rot.push_back(v0);
    rot.push_back(v1);
    rot.push_back(v2);
    // local rotation zeroed (main.cpp:131-136), translation kept
    cams.insert(cams.end(), {0.0, 0.0, 0.0, q[4], q[5], q[6]});
    nC++;
  }
  int nP = 0, per = -1;
  struct Ob { int cam; double x, y; };
  std::vector<Ob> obs;
  while (rc == PSBA_OK && read_data_line(fp, line)) {
    if (!parse_doubles(line, vals) || vals.size() < 4) { rc = PSBA_E_IO; break; }
    const int nfr = (int)vals[3];
    if (nfr < 0 || (double)nfr != vals[3]) { rc = PSBA_E_IO; break; }
    const int rest = (int)vals.size() - 4;
    if (per < 0) {  // covariance detection from the first line (readparams.cpp:272-283)
      if (nfr > 0 && rest == nfr * 7) per = 7;
      else if (nfr > 0 && rest == nfr * 6) per = 6;
      else per = 3;
    }
    if (rest != nfr * per) { rc = PSBA_E_IO; break; }
    obs.clear();
    for (int f = 0; f < nfr; f++) {
      const double *o = vals.data() + 4 + (size_t)per * f;
      const int cam = (int)o[0];
      if ((double)cam != o[0] || cam < 0 || cam >= nC) { rc = PSBA_E_IO; break; }
      obs.push_back({cam, o[1], o[2]});
    }
    if (rc != PSBA_OK) break;
    std::stable_sort(obs.begin(), obs.end(), [](const Ob &a, const Ob &b) { return a.cam < b.cam; });
    for (size_t f = 1; f < obs.size(); f++)
      if (obs[f].cam == obs[f - 1].cam) rc = PSBA_E_IO;  // a point seen twice by one camera
    pts.insert(pts.end(), {vals[0], vals[1], vals[2]});
    for (const Ob &o : obs) {
      iidx.push_back(nP);
      jidx.push_back(o.cam);
      impts.push_back(o.x);
      impts.push_back(o.y);
    }
    nP++;
  }
  fclose(fc);
  fclose(fp);
  if (rc == PSBA_OK && (nC == 0 || nP == 0 || iidx.empty())) rc = PSBA_E_IO;
  if (rc != PSBA_OK) return rc;
  out->nCams = nC;
  out->n3Dpts = nP;
  out->n2Dprojs = (int)iidx.size();
  out->Kparas = dup(K);
  out->impts = dup(impts);
  out->initrot = dup(rot);
  out->camsEx = dup(cams);
  out->pts3D = dup(pts);
  out->iidx = dup(iidx);
  out->jidx = dup(jidx);
  if (!out->Kparas || !out->impts || !out->initrot || !out->camsEx || !out->pts3D || !out->iidx ||
      !out->jidx) {
    psba_free_problem(out);
    return PSBA_E_NOMEM;
  }
  return PSBA_OK;
}

// ---- writer ------------------------------------------------------------------------------
// The reference declares printSBAMotionData / printSBAStructureData / printSBAData and keeps them
// commented out (PSBA/readparams.h:13-25); their output filter is vec2quat (PSBA/misc.cpp:60-85):
// the optimised local rotation v is turned back into a full quaternion.  Here the local rotation
// is composed onto initrot (q = q_l(v) (x) q0, the product the kernels use,
// CL_files/compute_exQT.cl:36-49), so the files written are again valid input: reading them back
// gives initrot = q and v = 0, i.e. the same cameras.  17 significant digits: doubles round-trip.
int psba_write_problem(const char *cams_file, const char *pts_file, int nCams, int n3Dpts, int n2Dprojs,
                       const double *Kparas, const double *initrot, const double *camsEx, const double *pts3D,
                       const double *impts, const int *iidx, const int *jidx, int with_K) {
  if (!cams_file || !pts_file || nCams <= 0 || n3Dpts <= 0 || n2Dprojs <= 0 || !initrot || !camsEx || !pts3D ||
      !impts || !iidx || !jidx || (with_K && !Kparas))
    return PSBA_E_INVALID;
  FILE *fc = fopen(cams_file, "w");
  if (!fc) return PSBA_E_IO;
  for (int j = 0; j < nCams; j++) {
    const double *v = camsEx + 6 * (size_t)j, *q0 = initrot + 4 * (size_t)j;
    const double sl = std::sqrt(1.0 - v[0] * v[0] - v[1] * v[1] - v[2] * v[2]);
    // Hamilton product (sl, v) (x) q0
    double q[4] = {sl * q0[0] - (v[0] * q0[1] + v[1] * q0[2] + v[2] * q0[3]),
                   sl * q0[1] + q0[0] * v[0] + (v[1] * q0[3] - v[2] * q0[2]),
                   sl * q0[2] + q0[0] * v[1] + (v[2] * q0[1] - v[0] * q0[3]),
                   sl * q0[3] + q0[0] * v[2] + (v[0] * q0[2] - v[1] * q0[1])};
    if (with_K)
      for (int k = 0; k < 5; k++) fprintf(fc, "%.17g ", Kparas[5 * (size_t)j + k]);
    fprintf(fc, "%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", q[0], q[1], q[2], q[3], v[3], v[4], v[5]);
  }
  if (fclose(fc) != 0) return PSBA_E_IO;
  FILE *fp = fopen(pts_file, "w");
  if (!fp) return PSBA_E_IO;
  int a = 0;
  for (int i = 0; i < n3Dpts; i++) {
    int b = a;
    while (b < n2Dprojs && iidx[b] == i) b++;
    fprintf(fp, "%.17g %.17g %.17g %d", pts3D[3 * (size_t)i], pts3D[3 * (size_t)i + 1], pts3D[3 * (size_t)i + 2], b - a);
    for (; a < b; a++) fprintf(fp, " %d %.17g %.17g", jidx[a], impts[2 * (size_t)a], impts[2 * (size_t)a + 1]);
    fprintf(fp, "\n");
  }
  const bool ok = a == n2Dprojs;  // observations must be point-major
  if (fclose(fp) != 0 || !ok) return ok ? PSBA_E_IO : PSBA_E_INVALID;
  return PSBA_OK;
}

// ---- Bundle-Adjustment-in-the-Large -> sba -------------------------------------------------
// BAL text: "ncams npts nobs", nobs lines "cam point x y", then 9 numbers per camera (Rodrigues
// vector r, translation t, focal f, radial k1 k2) and 3 per point, one number per line.  BAL's
// camera looks down -z and projects p = -P / P.z, x = f p (1 + k1 |p|^2 + k2 |p|^4); the
// reference's model looks down +z, divides by P.z and has no distortion.
// With F = diag(1, -1, -1) (a half turn about x): R' = F R, t' = F t gives P' = F P, so
// x' = f P'.x / P'.z equals BAL's x and y' equals minus BAL's y: the image y coordinates are
// negated, K = (f, 0, 0, 1, 0) -- the layout of the bundled BAL-derived files
// (data/Trafalgar-21-11315-cams.txt).  k1, k2 are dropped (the reference never applies
// distortion, PSBA/main.cpp:73,102-103); *max_abs_k (may be NULL) returns the largest |k1|, |k2|
// dropped so that the caller can judge it.  Observations are written point-major, cameras
// ascending, which is what every consumer of the format assumes.
int psba_convert_bal(const char *bal_file, const char *cams_out, const char *pts_out, double *max_abs_k) {
  if (!bal_file || !cams_out || !pts_out) return PSBA_E_INVALID;
  FILE *fb = fopen(bal_file, "r");
  if (!fb) return PSBA_E_IO;
  int nC = 0, nP = 0, nO = 0;
  if (fscanf(fb, "%d %d %d", &nC, &nP, &nO) != 3 || nC <= 0 || nP <= 0 || nO <= 0) {
    fclose(fb);
    return PSBA_E_IO;
  }
  struct Ob { int pt, cam; double x, y; };
  std::vector<Ob> obs((size_t)nO);
  for (auto &o : obs)
    if (fscanf(fb, "%d %d %lf %lf", &o.cam, &o.pt, &o.x, &o.y) != 4 || o.cam < 0 || o.cam >= nC || o.pt < 0 ||
        o.pt >= nP) {
      fclose(fb);
      return PSBA_E_IO;
    }
  std::vector<double> cam((size_t)9 * nC), pt((size_t)3 * nP);
  for (auto &v : cam)
    if (fscanf(fb, "%lf", &v) != 1) { fclose(fb); return PSBA_E_IO; }
  for (auto &v : pt)
    if (fscanf(fb, "%lf", &v) != 1) { fclose(fb); return PSBA_E_IO; }
  fclose(fb);
  std::stable_sort(obs.begin(), obs.end(), [](const Ob &a, const Ob &b) { return a.pt != b.pt ? a.pt < b.pt : a.cam < b.cam; });
  for (size_t k = 1; k < obs.size(); k++)
    if (obs[k].pt == obs[k - 1].pt && obs[k].cam == obs[k - 1].cam) return PSBA_E_IO;
  FILE *fc = fopen(cams_out, "w");
  if (!fc) return PSBA_E_IO;
  double kmax = 0.0;
  for (int j = 0; j < nC; j++) {
    const double *c = &cam[(size_t)9 * j];
    const double th = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    // Rodrigues vector -> unit quaternion; sin(th/2)/th -> 1/2 as th -> 0
    const double k = th > 1e-12 ? std::sin(0.5 * th) / th : 0.5;
    const double q[4] = {std::cos(0.5 * th), k * c[0], k * c[1], k * c[2]};
    // (0, 1, 0, 0) (x) q : the half turn about x applied after R
    const double qf[4] = {-q[1], q[0], -q[3], q[2]};
    fprintf(fc, "%.17g 0 0 1 0 %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", c[6], qf[0], qf[1], qf[2], qf[3], c[3],
            -c[4], -c[5]);
    kmax = std::max(kmax, std::max(std::fabs(c[7]), std::fabs(c[8])));
  }
  if (fclose(fc) != 0) return PSBA_E_IO;
  if (max_abs_k) *max_abs_k = kmax;
  FILE *fp = fopen(pts_out, "w");
  if (!fp) return PSBA_E_IO;
  size_t a = 0;
  for (int i = 0; i < nP; i++) {
    size_t b = a;
    while (b < obs.size() && obs[b].pt == i) b++;
    fprintf(fp, "%.17g %.17g %.17g %d", pt[(size_t)3 * i], pt[(size_t)3 * i + 1], pt[(size_t)3 * i + 2], (int)(b - a));
    for (; a < b; a++) fprintf(fp, " %d %.17g %.17g", obs[a].cam, obs[a].x, -obs[a].y);
    fprintf(fp, "\n");
  }
  if (fclose(fp) != 0) return PSBA_E_IO;
  return PSBA_OK;
}

}  // extern "C"
