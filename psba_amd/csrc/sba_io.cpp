// sba_io.cpp -- reader for the sba text format (host only).
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

}  // extern "C"
