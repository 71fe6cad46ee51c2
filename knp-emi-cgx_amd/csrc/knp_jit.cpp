// Run-time compilation of the membrane programs (the counterpart of the FFCx JIT in the reference: there every
// mechanism's UFL expression becomes generated C inside the facet kernels, KNPEMIx_problem.py:654-655).
//
// The bytecode programs uploaded with knp_set_program are turned into straight-line HIP functions knp_prog_<id>()
// (one statement per instruction, registers become locals), placed in front of csrc/knp_gamma_facets.inc -- the very
// source of the ahead-of-time facet kernel -- and compiled with hiprtc for the device's architecture.  The right-hand
// side assembly then launches the compiled kernel instead of the bytecode interpreter (3-10x faster on the membrane
// kernel; constants stay run-time data, so time-dependent constants need no recompilation).
// hiprtc is loaded with dlopen: without it (or with KNP_JIT=0, or on any compile error) the interpreter is used.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "knp_internal.hpp"

namespace {

struct Rtc {
    void* lib = nullptr;
    decltype(&hiprtcCreateProgram) create = nullptr;
    decltype(&hiprtcCompileProgram) compile = nullptr;
    decltype(&hiprtcGetCodeSize) code_size = nullptr;
    decltype(&hiprtcGetCode) code = nullptr;
    decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
    decltype(&hiprtcGetProgramLog) log = nullptr;
    decltype(&hiprtcDestroyProgram) destroy = nullptr;
    bool ok = false;
};

Rtc& rtc() {
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
#define LOAD(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, sym))
        LOAD(create, "hiprtcCreateProgram");
        LOAD(compile, "hiprtcCompileProgram");
        LOAD(code_size, "hiprtcGetCodeSize");
        LOAD(code, "hiprtcGetCode");
        LOAD(log_size, "hiprtcGetProgramLogSize");
        LOAD(log, "hiprtcGetProgramLog");
        LOAD(destroy, "hiprtcDestroyProgram");
#undef LOAD
        r.ok = r.create && r.compile && r.code_size && r.code && r.log_size && r.log && r.destroy;
    });
    return r;
}

std::string kernel_source_path() {
    Dl_info info;
    if (!dladdr(reinterpret_cast<void*>(&kernel_source_path), &info) || !info.dli_fname) return "";
    std::string so = info.dli_fname;                      // .../knp-emi-cgx_amd/cgx_hip/libknpemi_hip.so
    const size_t p = so.find_last_of('/');
    const std::string dir = p == std::string::npos ? "." : so.substr(0, p);
    return dir + "/../csrc/knp_gamma_facets.inc";
}

std::string reg(int r) { return "r" + std::to_string(r); }

// one statement per instruction; mirrors run_program() in knp_gamma_facets.inc case by case
bool emit_program(std::ostringstream& o, int id, const KnpProgram& p) {
    o << "__device__ __forceinline__ void knp_prog_" << id
      << "(const double* __restrict__ C, const double* ki, const double* ke, double phim, const double* aux, const double* xq, double* I) {\n";
    if (p.n_regs > 0) {
        o << "    double ";
        for (int r = 0; r < p.n_regs; ++r) o << (r ? ", " : "") << reg(r) << " = 0.0";
        o << ";\n";
    }
    for (int i = 0; i < p.n_instr; ++i) {
        const int op = p.h_code[4 * i], d = p.h_code[4 * i + 1], a = p.h_code[4 * i + 2], b = p.h_code[4 * i + 3];
        const std::string D = reg(d), A = reg(a), B = reg(b);
        o << "    ";
        switch (op) {
            case KNP_OP_CONST: o << D << " = C[" << a << "];"; break;
            case KNP_OP_KI: o << D << " = ki[" << a << "];"; break;
            case KNP_OP_KE: o << D << " = ke[" << a << "];"; break;
            case KNP_OP_PHIM: o << D << " = phim;"; break;
            case KNP_OP_AUX: o << D << " = aux[" << a << "];"; break;
            case KNP_OP_X: o << D << " = xq[" << a << "];"; break;
            case KNP_OP_ADD: o << D << " = " << A << " + " << B << ";"; break;
            case KNP_OP_SUB: o << D << " = " << A << " - " << B << ";"; break;
            case KNP_OP_MUL: o << D << " = " << A << " * " << B << ";"; break;
            case KNP_OP_DIV: o << D << " = " << A << " / " << B << ";"; break;
            case KNP_OP_NEG: o << D << " = -" << A << ";"; break;
            case KNP_OP_POW: o << D << " = pow(" << A << ", " << B << ");"; break;
            case KNP_OP_LN: o << D << " = log(" << A << ");"; break;
            case KNP_OP_EXP: o << D << " = exp(" << A << ");"; break;
            case KNP_OP_SQRT: o << D << " = sqrt(" << A << ");"; break;
            case KNP_OP_MAX: o << D << " = fmax(" << A << ", " << B << ");"; break;
            case KNP_OP_MIN: o << D << " = fmin(" << A << ", " << B << ");"; break;
            case KNP_OP_ABS: o << D << " = fabs(" << A << ");"; break;
            case KNP_OP_LT: o << D << " = " << A << " < " << B << " ? 1.0 : 0.0;"; break;
            case KNP_OP_GT: o << D << " = " << A << " > " << B << " ? 1.0 : 0.0;"; break;
            case KNP_OP_LE: o << D << " = " << A << " <= " << B << " ? 1.0 : 0.0;"; break;
            case KNP_OP_GE: o << D << " = " << A << " >= " << B << " ? 1.0 : 0.0;"; break;
            case KNP_OP_EQ: o << D << " = " << A << " == " << B << " ? 1.0 : 0.0;"; break;
            case KNP_OP_AND: o << D << " = (" << A << " != 0.0 && " << B << " != 0.0) ? 1.0 : 0.0;"; break;
            case KNP_OP_OR: o << D << " = (" << A << " != 0.0 || " << B << " != 0.0) ? 1.0 : 0.0;"; break;
            case KNP_OP_NOT: o << D << " = " << A << " != 0.0 ? 0.0 : 1.0;"; break;
            case KNP_OP_SEL: o << D << " = " << A << " != 0.0 ? " << B << " : " << D << ";"; break;
            case KNP_OP_OUT: o << "I[" << a << "] += " << B << ";"; break;
            case KNP_OP_MOV: o << D << " = " << A << ";"; break;
            case KNP_OP_POWI: o << D << " = powi_d(" << A << ", " << b << ");"; break;
            default: return false;
        }
        o << "\n";
    }
    o << "}\n\n";
    return true;
}

// compiled code objects, shared by every context of the process (keyed by source text + architecture)
std::map<std::string, std::vector<char>>& cache() {
    static std::map<std::string, std::vector<char>> c;
    return c;
}
std::mutex& cache_mutex() {
    static std::mutex m;
    return m;
}

}  // namespace

// whole translation unit for a set of programs: prelude, one function per program, dispatcher, the facet kernel source
static bool build_source(const std::vector<KnpProgram>& progs, std::string& out, std::string& err) {
    std::ifstream in(kernel_source_path());
    if (!in) { err = "kernel source " + kernel_source_path() + " not found"; return false; }
    std::stringstream kernel;
    kernel << in.rdbuf();
    std::ostringstream src;
    src << "#define KNP_GAMMA_JIT 1\n#define KNP_MAX_AUX " << KNP_MAX_AUX << "\n"
        << "typedef int knp_i32_t;\n#define int32_t knp_i32_t\n"
        << "struct DevParams { double dt, F, C_M, psi; double z[3], Di[3], De[3]; };\n"
        << "struct FieldPtrs { const double* ki[3]; const double* ke[3]; const double* phim; const double* aux[KNP_MAX_AUX]; };\n"
        << "__device__ __forceinline__ double powi_d(double x, int e);\n\n";
    for (size_t i = 0; i < progs.size(); ++i) {
        if (progs[i].h_code.size() != (size_t)4 * progs[i].n_instr) { err = "program code not retained"; return false; }
        if (!emit_program(src, (int)i, progs[i])) { err = "unknown opcode"; return false; }
    }
    src << "__device__ __forceinline__ void knp_jit_eval(int prog, const double* __restrict__ C, const double* ki, const double* ke, double phim,\n"
           "                                             const double* aux, const double* xq, double* I) {\n    switch (prog) {\n";
    for (size_t i = 0; i < progs.size(); ++i)
        src << "        case " << i << ": knp_prog_" << i << "(C, ki, ke, phim, aux, xq, I); break;\n";
    src << "        default: break;\n    }\n}\n\n" << kernel.str();
    out = src.str();
    return true;
}

// hiprtc compile (no device needed); code objects are cached per process by (architecture, source)
static bool compile_source(const std::string& text, const std::string& arch_opt, std::vector<char>& code, std::string& err) {
    Rtc& R = rtc();
    if (!R.ok) { err = "libhiprtc not available"; return false; }
    const std::string key = arch_opt + "\n" + text;
    {
        std::lock_guard<std::mutex> lock(cache_mutex());
        auto it = cache().find(key);
        if (it != cache().end()) { code = it->second; return true; }
    }
    hiprtcProgram prog = nullptr;
    if (R.create(&prog, text.c_str(), "knp_gamma_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { err = "hiprtcCreateProgram failed"; return false; }
    const char* opts[] = {arch_opt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off"};
    const hiprtcResult rc = R.compile(prog, 4, opts);
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        R.log_size(prog, &n);
        std::string log(n, '\0');
        if (n) R.log(prog, &log[0]);
        err = "hiprtc compile failed: " + log.substr(0, 2000);
        if (getenv("KNP_JIT_VERBOSE")) fprintf(stderr, "[knp jit] %s\n", err.c_str());
        R.destroy(&prog);
        return false;
    }
    size_t n = 0;
    R.code_size(prog, &n);
    code.resize(n);
    R.code(prog, code.data());
    R.destroy(&prog);
    std::lock_guard<std::mutex> lock(cache_mutex());
    cache()[key] = code;
    return true;
}

// Test hook (no device, no context): does this bytecode become a kernel for `arch` ("gfx950")?  0 on success; the
// generated source / compiler log is copied to `log`.
extern "C" int knp_jit_compile_check(const int32_t* code, int32_t n_instr, const char* arch, char* log, int32_t log_cap) {
    auto say = [&](const std::string& m) {
        if (log && log_cap > 0) { snprintf(log, (size_t)log_cap, "%s", m.c_str()); }
    };
    if (!code || n_instr < 0 || !arch) { say("bad arguments"); return KNP_E_ARG; }
    KnpProgram p;
    p.n_instr = n_instr;
    p.h_code.assign(code, code + (size_t)4 * n_instr);
    int max_reg = -1;
    for (int i = 0; i < n_instr; ++i)
        for (int k = 1; k < 4; ++k) max_reg = std::max(max_reg, code[4 * i + k] < KNP_MAX_PROG_REGS ? code[4 * i + k] : -1);
    p.n_regs = std::min(max_reg + 1, KNP_MAX_PROG_REGS);
    std::string src, err;
    std::vector<KnpProgram> progs(1, p);
    if (!build_source(progs, src, err)) { say(err); return KNP_E_STATE; }
    std::vector<char> obj;
    if (!compile_source(src, std::string("--offload-arch=") + arch, obj, err)) { say(err); return KNP_E_STATE; }
    say("ok: " + std::to_string(obj.size()) + " bytes of code object");
    return KNP_OK;
}

void knp_jit_release(knp_ctx* ctx) {
    if (ctx->jit_module) (void)hipModuleUnload((hipModule_t)ctx->jit_module);
    ctx->jit_module = nullptr;
    ctx->jit_fn[0] = ctx->jit_fn[1] = ctx->jit_fn[2] = nullptr;
}

// (re)build the native membrane kernel for the programs currently set; on any failure the interpreter stays in charge
void knp_jit_build(knp_ctx* ctx) {
    knp_jit_release(ctx);
    ctx->jit_msg.clear();
    const char* env = getenv("KNP_JIT");
    if (env && atoi(env) == 0) { ctx->jit_msg = "disabled by KNP_JIT=0"; return; }
    if (ctx->progs.empty()) return;
    std::string src, err;
    if (!build_source(ctx->progs, src, err)) { ctx->jit_msg = err; return; }
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { ctx->jit_msg = "no device properties"; return; }
    std::vector<char> code;
    if (!compile_source(src, std::string("--offload-arch=") + prop.gcnArchName, code, err)) { ctx->jit_msg = err; return; }
    hipModule_t mod = nullptr;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess) { (void)hipGetLastError(); ctx->jit_msg = "hipModuleLoadData failed"; return; }
    hipFunction_t f2 = nullptr, f3 = nullptr;
    if (hipModuleGetFunction(&f2, mod, "knp_gamma_vec_2d") != hipSuccess || hipModuleGetFunction(&f3, mod, "knp_gamma_vec_3d") != hipSuccess) {
        (void)hipGetLastError();
        (void)hipModuleUnload(mod);
        ctx->jit_msg = "compiled module lacks the kernels";
        return;
    }
    hipFunction_t f3m = nullptr;
    if (hipModuleGetFunction(&f3m, mod, "knp_gamma_vec_3d_many") != hipSuccess) { (void)hipGetLastError(); f3m = nullptr; }
    ctx->jit_module = mod;
    ctx->jit_fn[0] = f2;
    ctx->jit_fn[1] = f3;
    ctx->jit_fn[2] = f3m;
    ctx->jit_msg = "native";
}
