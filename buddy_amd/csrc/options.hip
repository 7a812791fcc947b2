// Per-handle options of the launchers -- the ONE place the BUDDY_* environment is read.
//
// Every A/B switch a launcher consults is a field of `Options`.  A network handle carries its own copy (Net::opt, changed with
// buddy_ncsnpp_set_option(handle, key, value)); while one of its calls runs, an OptScope makes that copy the calling thread's current options, which is
// what the launchers read (cur_opt()).  Outside a handle call -- the single-kernel entry points of the unit tests -- the process defaults apply.
// The process defaults come from the environment, parsed and VALIDATED once: a value outside its range, or a BUDDY_* name that is a NEAR MISS of a
// switch of this table (edit distance <= 2: BUDDY_UPCONVV, BUDDY_GN_FUSED), makes every handle creation (and buddy_option_check) fail with a message
// naming it -- a misspelt switch is never silently ignored.  A BUDDY_* name that resembles none of them (BUDDY_ROOT, BUDDY_DATA: the project is
// called BUDDy) belongs to somebody else: it is left alone, with one note on stderr.
#include "common.h"
#include "net.h"
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <algorithm>
#include <string>
#include <vector>

extern char** environ;

namespace buddy {
namespace {
struct Word { const char* name; int value; };
struct Entry {
  const char* key;      // buddy_ncsnpp_set_option key
  const char* env;      // environment variable of the process default
  int Options::*field;
  int lo, hi, dflt;
  const Word* words;    // symbolic values (environment only), terminated by {nullptr, 0}
};
const Word kConvWords[] = {{"", 0}, {"direct", 1}, {"wino2", 2}, {"wino4", 3}, {nullptr, 0}};
const Word kGemmWords[] = {{"fp32", 0}, {"bf16x3", 1}, {"f16x2", 2}, {nullptr, 0}};
const Word kAttnWords[] = {{"flash", 0}, {"bf16", 1}, {"f16", 2}, {"matrix", 3}, {"auto", 4}, {nullptr, 0}};
const Entry kTable[] = {
    {"conv", "BUDDY_CONV", &Options::conv, 0, 3, 0, kConvWords},
    {"gemm", "BUDDY_GEMM", &Options::gemm, 0, 2, 2, kGemmWords},
    {"attention", "BUDDY_ATTN", &Options::attn, 0, 4, 4, kAttnWords},
    {"gn_fuse", "BUDDY_GN_FUSE", &Options::gn_fuse, 0, 1, 1, nullptr},
    {"gn_fuse_bwdin", "BUDDY_GN_FUSE_BWDIN", &Options::gn_fuse_bwdin, 0, 1, 1, nullptr},
    {"gn_fuse_bwd", "BUDDY_GN_FUSE_BWD", &Options::gn_fuse_bwd, 0, 1, 1, nullptr},
    {"upconv", "BUDDY_UPCONV", &Options::upconv, 0, 1, 1, nullptr},
    {"c2_fuse", "BUDDY_C2_FUSE", &Options::c2_fuse, 0, 1, 1, nullptr},
    {"attn_tr", "BUDDY_ATTN_TR", &Options::attn_tr, 0, 1, 1, nullptr},
    {"attn_split", "BUDDY_ATTN_SPLIT", &Options::attn_split, 0, 4096, 0, nullptr},
    {"attn_nw", "BUDDY_ATTN_NW", &Options::attn_nw, 0, 8, 0, nullptr},
    {"igemm_epi", "BUDDY_IGEMM_EPI", &Options::igemm_epi, 0, 1, 1, nullptr},
    {"igemm_variant", "BUDDY_IGEMM_VARIANT", &Options::igemm_variant, 0, 2, 2, nullptr},
    {"wgemm_gen_epi", "BUDDY_WGEMM_GEN_EPI", &Options::wgemm_gen_epi, 0, 1, 1, nullptr},
    {"wgemm_xcdpos", "BUDDY_WGEMM_XCDPOS", &Options::wgemm_xcdpos, 0, 1, 1, nullptr},
    {"wgemm_epi", "BUDDY_WGEMM_EPI", &Options::wgemm_epi, 0, 1, 1, nullptr},
    {"wgemm_rt", "BUDDY_WGEMM_RT", &Options::wgemm_rt, 0, 2, 0, nullptr},
    {"wgemm_nt", "BUDDY_WGEMM_NT", &Options::wgemm_nt, 0, 3, 0, nullptr},
    {"gen_f16x2", "BUDDY_GEN_F16X2", &Options::gen_f16x2, 0, 2, 1, nullptr},
    {"gen_rows", "BUDDY_GEN_ROWS", &Options::gen_rows, 0, 64, 0, nullptr},
    {"gen_cp", "BUDDY_GEN_CP", &Options::gen_cp, 0, 2, 1, nullptr},
    {"gnb_nt", "BUDDY_GNB_NT", &Options::gnb_nt, 0, 1, 1, nullptr},
    {"wino_epi", "BUDDY_WINO_EPI", &Options::wino_epi, 0, 1, 1, nullptr},
    {"wino_abl", "BUDDY_WINO_ABL", &Options::wino_abl, 0, 3, 0, nullptr},
    {"wino_geo", "BUDDY_WINO_GEO", &Options::wino_geo, 42, 82, 42, nullptr},
    {"w6_xcd", "BUDDY_W6_XCD", &Options::w6_xcd, 0, 1, 1, nullptr},
    {"w6_nt", "BUDDY_W6_NT", &Options::w6_nt, 0, 1, 1, nullptr},
    {"gn_fast", "BUDDY_GN_FAST", &Options::gn_fast, 0, 1, 1, nullptr},
    {"ew_grid", "BUDDY_EW_GRID", &Options::ew_grid, 8, 22, 16, nullptr},
    {"gn_trips", "BUDDY_GN_TRIPS", &Options::gn_trips, 1, 64, 4, nullptr},
    {"c2in4", "BUDDY_C2IN4", &Options::c2in4, 0, 1, 1, nullptr},
    {"c2out_tiled", "BUDDY_C2OUT_TILED", &Options::c2out_tiled, 0, 2, 2, nullptr},
    {"fir_lds", "BUDDY_FIR_LDS", &Options::fir_lds, 0, 1, 1, nullptr},
    {"op_graph", "BUDDY_OP_GRAPH", &Options::op_graph, 0, 1, 1, nullptr},
};
// environment variables that are not launcher options (read elsewhere: the profiling dump path below, the bench driver)
const char* const kOtherEnv[] = {"BUDDY_PROF_DUMP", "BUDDY_BENCH_PROF"};

struct Defaults { Options opt; std::string error; };
// Levenshtein distance, capped: names of a few tens of characters, called once per unknown variable
int edit_distance(const std::string& a, const std::string& b) {
  std::vector<int> prev(b.size() + 1), cur(b.size() + 1);
  for (size_t j = 0; j <= b.size(); ++j) prev[j] = (int)j;
  for (size_t i = 1; i <= a.size(); ++i) {
    cur[0] = (int)i;
    for (size_t j = 1; j <= b.size(); ++j) cur[j] = std::min(std::min(prev[j] + 1, cur[j - 1] + 1), prev[j - 1] + (a[i - 1] != b[j - 1]));
    prev.swap(cur);
  }
  return prev[b.size()];
}
bool parse_value(const Entry& e, const char* text, int* out) {
  if (e.words)
    for (const Word* w = e.words; w->name; ++w)
      if (!strcmp(w->name, text)) { *out = w->value; return true; }
  char* end = nullptr;
  const long v = strtol(text, &end, 10);
  if (end == text || *end != '\0' || v < e.lo || v > e.hi) return false;
  *out = (int)v;
  return true;
}
const Defaults& defaults() {
  static const Defaults d = [] {
    Defaults r;
    for (const Entry& e : kTable) r.opt.*(e.field) = e.dflt;
    for (char** ev = environ; ev && *ev; ++ev) {
      if (strncmp(*ev, "BUDDY_", 6)) continue;
      const char* eq = strchr(*ev, '=');
      if (!eq) continue;
      const std::string name(*ev, eq - *ev);
      bool known = false;
      for (const char* o : kOtherEnv) known = known || name == o;
      for (const Entry& e : kTable) {
        if (name != e.env) continue;
        known = true;
        int v;
        if (!parse_value(e, eq + 1, &v)) { if (r.error.empty()) r.error = "environment: bad value '" + std::string(eq + 1) + "' for " + name; }
        else r.opt.*(e.field) = v;
      }
      if (known) continue;
      const char* near = nullptr;
      for (const Entry& e : kTable) if (edit_distance(name, e.env) <= 2) near = e.env;
      for (const char* o : kOtherEnv) if (edit_distance(name, o) <= 2) near = o;
      if (near) { if (r.error.empty()) r.error = "environment: unknown variable " + name + " (did you mean " + near + "?  the BUDDY_* switches are listed in csrc/options.hip)"; }
      else fprintf(stderr, "libbuddy_hip: note: environment variable %s is not one of this library's switches (csrc/options.hip); ignored\n", name.c_str());
    }
    return r;
  }();
  return d;
}
thread_local const Options* tl_cur = nullptr;
}  // namespace

const Options& default_options() { return defaults().opt; }
int options_check() {
  if (!defaults().error.empty()) { set_error(defaults().error); return BUDDY_ERR_ARG; }
  return BUDDY_OK;
}
const Options& cur_opt() { return tl_cur ? *tl_cur : defaults().opt; }
OptScope::OptScope(const Options* o) : prev(tl_cur) { tl_cur = o; }
OptScope::~OptScope() { tl_cur = prev; }
int option_set(Options& o, const char* key, int value) {
  if (!key) { set_error("null option key"); return BUDDY_ERR_ARG; }
  for (const Entry& e : kTable)
    if (!strcmp(e.key, key)) {
      if (value < e.lo || value > e.hi) { set_error(std::string("option ") + key + ": value out of range"); return BUDDY_ERR_ARG; }
      o.*(e.field) = value;
      return BUDDY_OK;
    }
  set_error(std::string("unknown option '") + key + "'");
  return BUDDY_ERR_ARG;
}
int option_get(const Options& o, const char* key, int* value) {
  if (!key || !value) { set_error("null argument"); return BUDDY_ERR_ARG; }
  for (const Entry& e : kTable)
    if (!strcmp(e.key, key)) { *value = o.*(e.field); return BUDDY_OK; }
  set_error(std::string("unknown option '") + key + "'");
  return BUDDY_ERR_ARG;
}
const char* prof_dump_path() { static const char* p = getenv("BUDDY_PROF_DUMP"); return p; }

}  // namespace buddy
