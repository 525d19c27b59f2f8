// mcx_user.hip -- MCX_VL_SOURCE: a user's likelihood, given as HIP source of device functions, compiled INTO the
// step kernels at run time.  The reference's plug-in surface is one virtual call per step on the whole batch
// (src/vlfunc.hh:9-12, called at src/mcpar.cc:60,160); its GPU counterpart with a separately compiled user kernel
// (MCX_VL_DEVICE) costs three launches per step with the chain state round-tripping HBM in between.  Here the user's
// functions become Lik<LIK_USER, LPC> of mcx_device.hpp: burn-in with its tuner, main loop, Welford, sample emission stay
// ONE launch per segment, state in registers, exactly as for the built-in likelihoods.
//
// hiprtc is loaded with dlopen on first use (no link dependency); the kernel headers travel inside libmcx.so as
// strings (mcx_rtc_headers.inc, made by tools/embed_headers.py).  Code objects are cached per (source, lanes per
// chain) for the life of the process.
#include "mcx_engine_internal.hpp"

#include <iterator>
#include <map>
#include <memory>
#include <mutex>

#include "mcx_rtc_headers.inc"

namespace {

typedef struct _hiprtcProgram *rtcProgram;
struct RtcApi {
  void *handle = nullptr;
  int (*CreateProgram)(rtcProgram *, const char *, const char *, int, const char **, const char **) = nullptr;
  int (*CompileProgram)(rtcProgram, int, const char **) = nullptr;
  int (*GetProgramLogSize)(rtcProgram, size_t *) = nullptr;
  int (*GetProgramLog)(rtcProgram, char *) = nullptr;
  int (*GetCodeSize)(rtcProgram, size_t *) = nullptr;
  int (*GetCode)(rtcProgram, char *) = nullptr;
  int (*DestroyProgram)(rtcProgram *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string why;
};
RtcApi g_rtc;
std::mutex g_user_m;

bool rtc_load()
{
  if (g_rtc.handle) return true;
  if (!g_rtc.why.empty()) return false;
  const char *names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
  void *h = nullptr;
  for (const char *n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!h) {
    const char *de = dlerror();
    g_rtc.why = std::string("libhiprtc.so not loadable: ") + (de ? de : "?");
    return false;
  }
  bool ok = true;
  auto sym = [&](const char *n) { void *p = dlsym(h, n); if (!p) { ok = false; g_rtc.why = std::string("missing hiprtc symbol ") + n; } return p; };
  g_rtc.CreateProgram = (decltype(g_rtc.CreateProgram))sym("hiprtcCreateProgram");
  g_rtc.CompileProgram = (decltype(g_rtc.CompileProgram))sym("hiprtcCompileProgram");
  g_rtc.GetProgramLogSize = (decltype(g_rtc.GetProgramLogSize))sym("hiprtcGetProgramLogSize");
  g_rtc.GetProgramLog = (decltype(g_rtc.GetProgramLog))sym("hiprtcGetProgramLog");
  g_rtc.GetCodeSize = (decltype(g_rtc.GetCodeSize))sym("hiprtcGetCodeSize");
  g_rtc.GetCode = (decltype(g_rtc.GetCode))sym("hiprtcGetCode");
  g_rtc.DestroyProgram = (decltype(g_rtc.DestroyProgram))sym("hiprtcDestroyProgram");
  g_rtc.GetErrorString = (decltype(g_rtc.GetErrorString))sym("hiprtcGetErrorString");
  if (!ok) return false;
  g_rtc.handle = h;
  return true;
}

// the flags libmcx.so itself is built with (Makefile HIPFLAGS): the MCX arithmetic is a contract on bits
const char *const RTC_FLAGS[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                                 "-fno-gpu-flush-denormals-to-zero", "-Wno-unused-function"};

// source -> code object; log = the compiler's words on failure
int rtc_compile(const std::string &src, const char *name, const std::vector<std::string> &defs, std::vector<char> &code, std::string &log)
{
  if (!rtc_load()) return fail(MCX_ERR_UNSUPPORTED, "run-time compilation unavailable: %s", g_rtc.why.c_str());
  // (a user's text may well start with #include <hip/hip_runtime.h>: hiprtc has the runtime's declarations built in and no
  // such file, so an empty one stands in)
  const char *hn[] = {"mcx_numerics.hpp", "mcx_device.hpp", "mcx_fastb.hpp", "mcx_persist.hpp", "hip/hip_runtime.h"};
  const char *hs[] = {k_hdr_mcx_numerics, k_hdr_mcx_device, k_hdr_mcx_fastb, k_hdr_mcx_persist,
                      "// the HIP runtime declarations are built into hiprtc\n"};
  rtcProgram prog = nullptr;
  int r = g_rtc.CreateProgram(&prog, src.c_str(), name, 5, hs, hn);
  if (r != 0) return fail(MCX_ERR_HIP, "hiprtcCreateProgram: %s", g_rtc.GetErrorString(r));
  std::vector<const char *> opts(std::begin(RTC_FLAGS), std::end(RTC_FLAGS));
  for (const std::string &d : defs) opts.push_back(d.c_str());
  r = g_rtc.CompileProgram(prog, (int)opts.size(), opts.data());
  size_t nlog = 0;
  if (g_rtc.GetProgramLogSize(prog, &nlog) == 0 && nlog > 1) {
    log.resize(nlog);
    (void)g_rtc.GetProgramLog(prog, &log[0]);
    while (!log.empty() && (log.back() == '\0' || log.back() == '\n')) log.pop_back();
  }
  if (r != 0) {
    (void)g_rtc.DestroyProgram(&prog);
    // the tail is where clang puts the error count; the head is where the first error is: keep both ends
    std::string shown = log.size() > 3000 ? log.substr(0, 2200) + "\n[...]\n" + log.substr(log.size() - 600) : log;
    return fail(MCX_ERR_VLFUNC, "the likelihood source does not compile (%s):\n%s", g_rtc.GetErrorString(r), shown.c_str());
  }
  size_t nb = 0;
  r = g_rtc.GetCodeSize(prog, &nb);
  if (r == 0) {
    code.resize(nb);
    r = g_rtc.GetCode(prog, code.data());
  }
  (void)g_rtc.DestroyProgram(&prog);
  if (r != 0) return fail(MCX_ERR_HIP, "hiprtcGetCode: %s", g_rtc.GetErrorString(r));
  return MCX_OK;
}

// What the run-time translation unit looks like around the user's text.  The user's functions live in the global
// namespace and may use mcx_numerics.hpp (mcx::logf_v1, mcx::expf_v2, ... -- the functions the built-ins and the CPU
// oracle use, for results that agree with them bit for bit).
const char *const TU_HEAD =
    "#include \"mcx_numerics.hpp\"\n"
    "#line 1 \"mcx_user_likelihood\"\n";
const char *const TU_TAIL =
    "\n#line 1 \"mcx_user_kernels\"\n"
    "#ifdef MCX_USER_BLOCK_FORM\n"
    "#define MCX_USER_LIK 1\n"
    "#ifndef MCX_USER_FINISH\n"
    "__device__ __forceinline__ float mcx_user_finish(float s, int, const float *) { return s; }\n"
    "#endif\n"
    "#else\n"
    "#define MCX_USER_LIK 2\n"
    "#endif\n"
    "#include \"mcx_fastb.hpp\"\n"
    "using namespace mcx;\n"
    "#define MCX_USER_KERNEL extern \"C\" __global__ __launch_bounds__(BLOCK) void\n"
    "#if MCX_USER_LIK == 1\n"
    "extern \"C\" __global__ void mcx_user_is_block_form() {}\n"  // (the host asks the module which form the text has)
    "#endif\n"
    // whole-vector form: as many blocks per lane as the hot-path kernel has (4, or 2 for np <= 8) -- a chain of np <= 16 is
    // then ONE lane and the function is evaluated once per chain, not once per lane of it
    "#if MCX_USER_LIK == 2 && MCX_USER_LPC >= 2 && MCX_USER_LPC <= 8\n"
    "#define MCX_USER_BPL (MCX_USER_LPC >= 4 ? 4 : 2)\n"
    "MCX_USER_KERNEL mcx_user_fastb_burn(const SegArgs a) { const uint32_t w = fused_fastb_body<MCX_USER_LPC / MCX_USER_BPL, MCX_USER_BPL, false, LIK_USER>(a); tuner_epilogue(a, w); }\n"
    "MCX_USER_KERNEL mcx_user_fastb_main(const SegArgs a) { const uint32_t w = fused_fastb_body<MCX_USER_LPC / MCX_USER_BPL, MCX_USER_BPL, true, LIK_USER>(a); tuner_epilogue(a, w); }\n"
    "#endif\n"
    "#if MCX_USER_LPC <= 8\n"
    "MCX_USER_KERNEL mcx_user_fast_burn(const SegArgs a) { const uint32_t w = fused_fast_body<MCX_USER_LPC, false, LIK_USER, false, false>(a); tuner_epilogue(a, w); }\n"
    "MCX_USER_KERNEL mcx_user_fast_main(const SegArgs a) { const uint32_t w = fused_fast_body<MCX_USER_LPC, true, LIK_USER, false, false>(a); tuner_epilogue(a, w); }\n"
    "MCX_USER_KERNEL mcx_user_full_burn(const SegArgs a) { const uint32_t w = fused_fast_body<MCX_USER_LPC, false, LIK_USER, false, true>(a); tuner_epilogue(a, w); }\n"
    "MCX_USER_KERNEL mcx_user_full_main(const SegArgs a) { const uint32_t w = fused_fast_body<MCX_USER_LPC, true, LIK_USER, false, true>(a); tuner_epilogue(a, w); }\n"
    "#endif\n"
    "MCX_USER_KERNEL mcx_user_steps_burn(const SegArgs a) { fused_steps_body<MCX_USER_LPC, LIK_USER, false>(a); }\n"
    "MCX_USER_KERNEL mcx_user_steps_main(const SegArgs a) { fused_steps_body<MCX_USER_LPC, LIK_USER, true>(a); }\n"
    "MCX_USER_KERNEL mcx_user_eval(const float *x, float *y, int n, int d, const float *lik, int ncomp, int vec4) { eval_body<MCX_USER_LPC, LIK_USER>(x, y, n, d, lik, ncomp, vec4); }\n";

// The one-launch small-n kernel (mcx_persist.hpp) around the same text: compiled on demand, one instantiation per
// (lanes per chain, blocks per lane, recorders) the engine picks for the run -- a k_run_small is the largest kernel of the
// library (2-4 s of hiprtc each).  Block form only: the owners' dependent chain has no room for an LDS round trip.
const char *const TU_TAIL_SMALL =
    "\n#line 1 \"mcx_user_small_kernel\"\n"
    "#ifndef MCX_USER_BLOCK_FORM\n"
    "#error the one-launch small-n kernel takes the block form only\n"
    "#endif\n"
    "#define MCX_USER_LIK 1\n"
    "#ifndef MCX_USER_FINISH\n"
    "__device__ __forceinline__ float mcx_user_finish(float s, int, const float *) { return s; }\n"
    "#endif\n"
    "#include \"mcx_persist.hpp\"\n"
    "using namespace mcx;\n"
    "extern \"C\" __global__ __launch_bounds__(PBLOCK) void mcx_user_small(const RunArgs a)\n"
    "{ run_small_body<MCX_USER_LPC2, MCX_USER_BPL, LIK_USER, (MCX_USER_REC != 0)>(a); }\n";

}  // namespace

struct UserLik {
  std::string source;
  int lpc = 0;
  int device = -1;
  hipModule_t mod = nullptr;
  hipFunction_t fast[2] = {nullptr, nullptr}, full[2] = {nullptr, nullptr}, steps[2] = {nullptr, nullptr}, eval = nullptr;
  hipFunction_t fastb[2] = {nullptr, nullptr};  // whole-vector form, 2 <= lpc <= 8: the hot-path kernel with `bpl` blocks per lane
  int bpl = 1;
  bool block_form = false;
  int np = 0;
  struct Small { hipModule_t mod = nullptr; hipFunction_t fn = nullptr; long long resident = -1; size_t lds = 0; };
  std::map<int, Small> small;  // by bpl * 2 + rec: the one-launch small-n kernel, compiled when a run first wants it
  double compile_ms = 0.0;
  ~UserLik()
  {
    if (mod) (void)hipModuleUnload(mod);
    for (auto &kv : small)
      if (kv.second.mod) (void)hipModuleUnload(kv.second.mod);
  }
};

namespace {
// never destroyed: at process exit the HIP runtime may be gone before a static destructor could unload a module
std::map<std::string, std::shared_ptr<UserLik>> &g_user_cache = *new std::map<std::string, std::shared_ptr<UserLik>>;
}

int user_lik_get(const char *source, int np, std::shared_ptr<UserLik> *out)
{
  const int lpc = lpc_for(np);
  if (!source || !*source) return fail(MCX_ERR_VLFUNC, "MCX_VL_SOURCE without source text (ctx)");
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_user_m);
  const std::string key = std::to_string(dev) + ":" + std::to_string(np) + ":" + std::to_string(std::hash<std::string>{}(source)) + ":" +
                          std::to_string(std::strlen(source));
  auto it = g_user_cache.find(key);
  if (it != g_user_cache.end() && it->second->source == source) {
    *out = it->second;
    return MCX_OK;
  }
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<char> code;
  std::string log;
  MCXCHK(rtc_compile(std::string(TU_HEAD) + source + TU_TAIL, "mcx_user.hip",
                     {"-DMCX_USER_LPC=" + std::to_string(lpc), "-DMCX_USER_NP=" + std::to_string(np)}, code, log));
  auto u = std::make_shared<UserLik>();
  u->source = source;
  u->lpc = lpc;
  u->device = dev;
  HIPCHK(hipModuleLoadData(&u->mod, code.data()));
  if (lpc <= 8) {
    HIPCHK(hipModuleGetFunction(&u->fast[0], u->mod, "mcx_user_fast_burn"));
    HIPCHK(hipModuleGetFunction(&u->fast[1], u->mod, "mcx_user_fast_main"));
    HIPCHK(hipModuleGetFunction(&u->full[0], u->mod, "mcx_user_full_burn"));
    HIPCHK(hipModuleGetFunction(&u->full[1], u->mod, "mcx_user_full_main"));
  }
  if (lpc >= 2 && lpc <= 8 && hipModuleGetFunction(&u->fastb[0], u->mod, "mcx_user_fastb_burn") == hipSuccess &&
      hipModuleGetFunction(&u->fastb[1], u->mod, "mcx_user_fastb_main") == hipSuccess) {
    u->bpl = lpc >= 4 ? 4 : 2;
  } else {
    (void)hipGetLastError();  // (block form: the kernels are not in the module)
    u->fastb[0] = u->fastb[1] = nullptr;
  }
  {
    hipFunction_t marker = nullptr;
    u->block_form = hipModuleGetFunction(&marker, u->mod, "mcx_user_is_block_form") == hipSuccess;
    if (!u->block_form) (void)hipGetLastError();
  }
  u->np = np;
  HIPCHK(hipModuleGetFunction(&u->steps[0], u->mod, "mcx_user_steps_burn"));
  HIPCHK(hipModuleGetFunction(&u->steps[1], u->mod, "mcx_user_steps_main"));
  HIPCHK(hipModuleGetFunction(&u->eval, u->mod, "mcx_user_eval"));
  u->compile_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (getenv("MCX_VERBOSE"))
    fprintf(stderr, "mcx: user likelihood compiled for %d lanes per chain in %.0f ms (%zu bytes of code)\n", lpc, u->compile_ms, code.size());
  // (a host that writes its constants into the text instead of `par` makes a new entry per value: beyond 64 entries
  // the ones no engine holds any more are let go, modules and all)
  size_t cache_max = 64;
  if (const char *cm = getenv("MCX_USER_CACHE_MAX")) cache_max = (size_t)std::max(1L, atol(cm));  // (tests)
  if (g_user_cache.size() >= cache_max)
    for (auto c = g_user_cache.begin(); c != g_user_cache.end();) c = c->second.use_count() == 1 ? g_user_cache.erase(c) : std::next(c);
  g_user_cache[key] = u;
  *out = u;
  return MCX_OK;
}

// which of the user's step kernels a segment takes: the same tests as launch_fused_plain for the built-ins
int user_lik_variant(int lpc, const mcx::SegArgs &a)
{
  if (lpc <= 8 && a.vec4 && !a.mask) return a.diag ? 0 : 1;  // hot-path kernel, diagonal / full factor
  return 2;                                                   // generic kernel: any d, accept mask
}

int user_lik_launch_fused(const UserLik &u, bool main, const mcx::SegArgs &a, hipStream_t st)
{
  const int v = user_lik_variant(u.lpc, a);
  hipFunction_t f = v == 0 ? u.fast[main ? 1 : 0] : (v == 1 ? u.full[main ? 1 : 0] : u.steps[main ? 1 : 0]);
  int lanes = u.lpc;
  if (v == 0 && u.fastb[0]) {  // whole-vector form on the diagonal hot path: several blocks per lane
    f = u.fastb[main ? 1 : 0];
    lanes = u.lpc / u.bpl;
  }
  if (!f) return fail(MCX_ERR_UNSUPPORTED, "no user step kernel for this configuration");
  mcx::SegArgs arg = a;
  void *args[] = {&arg};
  const unsigned grid = (unsigned)(((size_t)a.n * lanes + mcx::BLOCK - 1) / mcx::BLOCK);
  HIPCHK(hipModuleLaunchKernel(f, grid, 1, 1, mcx::BLOCK, 1, 1, 0, st, args, nullptr));
  return MCX_OK;
}

int user_lik_launch_eval(const UserLik &u, const float *x, float *y, int n, int d, const float *par, int ncomp, hipStream_t st)
{
  int vec4 = (d % 4 == 0) ? 1 : 0;
  void *args[] = {&x, &y, &n, &d, &par, &ncomp, &vec4};
  const unsigned grid = (unsigned)(((size_t)n * u.lpc + mcx::BLOCK - 1) / mcx::BLOCK);
  HIPCHK(hipModuleLaunchKernel(u.eval, grid, 1, 1, mcx::BLOCK, 1, 1, 0, st, args, nullptr));
  return MCX_OK;
}

double user_lik_compile_ms(const UserLik &u) { return u.compile_ms; }

// may a run with this likelihood take the one-launch small-n kernel?
bool user_lik_small_ok(const UserLik &u) { return u.block_form && u.lpc <= 8; }

// mcxk_launch_persist for LIK_USER: the kernel for (blocks per lane, recorders) is built on first use
hipError_t user_lik_launch_small(UserLik &u, int bpl, const mcx::RunArgs &a, hipStream_t st)
{
  const int lpc2 = u.lpc / bpl, rec = mcxk_persist_recorders(a.own, bpl) ? 1 : 0;
  UserLik::Small *k = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_user_m);
    k = &u.small[bpl * 2 + rec];
    if (!k->fn) {
      const auto t0 = std::chrono::steady_clock::now();
      std::vector<char> code;
      std::string log;
      if (rtc_compile(std::string(TU_HEAD) + u.source + TU_TAIL_SMALL, "mcx_user_small.hip",
                      {"-DMCX_USER_LPC=" + std::to_string(u.lpc), "-DMCX_USER_NP=" + std::to_string(u.np),
                       "-DMCX_USER_LPC2=" + std::to_string(lpc2), "-DMCX_USER_BPL=" + std::to_string(bpl),
                       "-DMCX_USER_REC=" + std::to_string(rec)}, code, log) != MCX_OK)
        return hipErrorInvalidValue;  // (mcx_last_error holds the compiler's words)
      hipError_t he = hipModuleLoadData(&k->mod, code.data());
      if (he == hipSuccess) he = hipModuleGetFunction(&k->fn, k->mod, "mcx_user_small");
      if (he != hipSuccess) return he;
      if (getenv("MCX_VERBOSE"))
        fprintf(stderr, "mcx: user likelihood compiled into the one-launch kernel (%d lanes per chain, %d blocks per lane%s) in %.0f ms\n",
                lpc2, bpl, rec ? ", recorders" : "", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
  }
  const unsigned nwg = (unsigned)((a.nown + a.own - 1) / a.own);
  const size_t lds = mcxk_persist_lds_bytes(lpc2, bpl, a.own);
  if (a.nburn > 0) {  // tuner meetings inside: every workgroup must be resident at once (mcx_k_persist.hip: go2)
    std::lock_guard<std::mutex> lk(g_user_m);
    if (k->resident < 0 || k->lds != lds) {
      int per_cu = 0, ncu = 0, dev = 0;
      hipError_t he = hipGetDevice(&dev);
      if (he == hipSuccess) he = hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k->fn, mcx::PBLOCK, lds);
      if (he == hipSuccess) he = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
      if (he != hipSuccess) return he;
      k->resident = (long long)per_cu * ncu;
      k->lds = lds;
    }
    if (k->resident < (long long)nwg) return hipErrorCooperativeLaunchTooLarge;
  }
  mcx::RunArgs arg = a;
  void *args[] = {&arg};
  return hipModuleLaunchKernel(k->fn, nwg, 1, 1, mcx::PBLOCK, 1, 1, (unsigned)lds, st, args, nullptr);
}

extern "C" int mcx_user_source_available(void)
{
  std::lock_guard<std::mutex> lk(g_user_m);
  if (rtc_load()) return 1;
  (void)fail(MCX_ERR_UNSUPPORTED, "%s", g_rtc.why.c_str());
  return 0;
}

// The compile step alone (no GPU needed: hiprtc cross-compiles for gfx950 like hipcc): does this text build into the
// step kernels for a chain of `np` parameters?  -> bytes of the code object; MCX_ERR_VLFUNC + the compiler's messages if not.
extern "C" int mcx_debug_user_source_compile(const char *source, int np, size_t *code_bytes)
{
  if (!source || np < 1 || np > mcx::MAXD) return fail(MCX_ERR_INVALID, "bad arguments");
  std::lock_guard<std::mutex> lk(g_user_m);
  std::vector<char> code;
  std::string log;
  MCXCHK(rtc_compile(std::string(TU_HEAD) + source + TU_TAIL, "mcx_user.hip",
                     {"-DMCX_USER_LPC=" + std::to_string(lpc_for(np)), "-DMCX_USER_NP=" + std::to_string(np)}, code, log));
  if (code_bytes) *code_bytes = code.size();
  return MCX_OK;
}

// the same for the one-launch small-n kernel around a block-form text (what user_lik_launch_small builds on first use)
extern "C" int mcx_debug_user_source_compile_small(const char *source, int np, int bpl, int rec, size_t *code_bytes)
{
  if (!source || np < 1 || np > 32 || (bpl != 1 && bpl != 2) || lpc_for(np) % bpl) return fail(MCX_ERR_INVALID, "bad arguments");
  std::lock_guard<std::mutex> lk(g_user_m);
  std::vector<char> code;
  std::string log;
  const int lpc = lpc_for(np);
  MCXCHK(rtc_compile(std::string(TU_HEAD) + source + TU_TAIL_SMALL, "mcx_user_small.hip",
                     {"-DMCX_USER_LPC=" + std::to_string(lpc), "-DMCX_USER_NP=" + std::to_string(np), "-DMCX_USER_LPC2=" + std::to_string(lpc / bpl),
                      "-DMCX_USER_BPL=" + std::to_string(bpl), "-DMCX_USER_REC=" + std::to_string(rec ? 1 : 0)}, code, log));
  if (code_bytes) *code_bytes = code.size();
  return MCX_OK;
}

// A whole user KERNEL from source (the MCX_VL_DEVICE contract: extern "C" __global__ void f(int npset, const float *x,
// float *y)) for hosts that have no hipcc at hand: -> hipFunction_t for mcx_vlfunc.ctx.  The module lives until the
// process ends.
extern "C" int mcx_user_kernel_compile(const char *source, const char *symbol, void **function)
{
  if (!source || !symbol || !function) return fail(MCX_ERR_INVALID, "bad arguments");
  MCXCHK(need_device());
  std::lock_guard<std::mutex> lk(g_user_m);
  std::vector<char> code;
  std::string log;
  MCXCHK(rtc_compile(std::string(TU_HEAD) + source, "mcx_user_kernel.hip", {}, code, log));
  hipModule_t mod = nullptr;
  hipFunction_t f = nullptr;
  HIPCHK(hipModuleLoadData(&mod, code.data()));
  const hipError_t ge = hipModuleGetFunction(&f, mod, symbol);
  if (ge != hipSuccess) {
    (void)hipGetLastError();  // (the runtime remembers the failure: the next hipGetLastError() of this thread must not find it)
    (void)hipModuleUnload(mod);
    return fail(MCX_ERR_VLFUNC, "kernel '%s' not found in the compiled source (declare it extern \"C\" __global__)", symbol);
  }
  *function = (void *)f;
  return MCX_OK;
}
