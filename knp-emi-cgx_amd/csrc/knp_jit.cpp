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

constexpr int KNP_JIT_UNIFORMS = 16;   // values a program may compute once per thread instead of once per quadrature point

// one statement of straight-line code for instruction (op, d, a, b); false: unknown opcode
bool emit_statement(std::ostringstream& o, int op, int d, int a, int b) {
    const std::string D = reg(d), A = reg(a), B = reg(b);
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
    return true;
}

// which source registers an instruction reads (SEL also reads its destination)
void sources(int op, int d, int a, int b, int out[3], int& n) {
    n = 0;
    switch (op) {
        case KNP_OP_CONST: case KNP_OP_KI: case KNP_OP_KE: case KNP_OP_PHIM: case KNP_OP_AUX: case KNP_OP_X: break;
        case KNP_OP_NEG: case KNP_OP_LN: case KNP_OP_EXP: case KNP_OP_SQRT: case KNP_OP_ABS: case KNP_OP_NOT: case KNP_OP_MOV: case KNP_OP_POWI:
            out[n++] = a; break;
        case KNP_OP_OUT: out[n++] = b; break;
        case KNP_OP_SEL: out[n++] = a; out[n++] = b; out[n++] = d; break;
        default: out[n++] = a; out[n++] = b; break;
    }
}

// Two functions per program, one statement per instruction (mirrors run_program() in knp_gamma_facets.inc case by case):
//   knp_prog_<id>_pre(C, U)  -- the instructions whose operands trace back to the constant table only (psi / z_k, exp(-t / a_syn), ...:
//                               a division or an exponential each, the same for every quadrature point), executed once per thread;
//   knp_prog_<id>(C, U, ...) -- the program per quadrature point, where such an instruction is a read of U.
// The compiler does not hoist them by itself out of the unrolled point loop (every copy re-did its four divisions and the exponential).
bool emit_program(std::ostringstream& o, int id, const KnpProgram& p) {
    const int nr = std::max(p.n_regs, 1);
    std::vector<char> uni((size_t)nr, 0);          // register currently holds a constant-only value
    std::vector<int> slot((size_t)p.n_instr, -1);  // instruction -> index into U (computed instructions only; CONST loads stay loads)
    int n_slots = 0;
    for (int i = 0; i < p.n_instr; ++i) {
        const int op = p.h_code[4 * i], d = p.h_code[4 * i + 1], a = p.h_code[4 * i + 2], b = p.h_code[4 * i + 3];
        int src[3], ns = 0;
        sources(op, d, a, b, src, ns);
        if (op == KNP_OP_OUT) continue;
        bool u = op == KNP_OP_CONST;
        if (!u && ns > 0) {
            u = true;
            for (int k = 0; k < ns; ++k) u = u && src[k] >= 0 && src[k] < nr && uni[(size_t)src[k]];
            if (u) {
                if (n_slots < KNP_JIT_UNIFORMS) slot[(size_t)i] = n_slots++;
                else u = false;                       // table full: computed per point as before (operands are still available there)
            }
        }
        if (d >= 0 && d < nr) uni[(size_t)d] = u;
    }
    auto declare = [&](std::ostringstream& s) {
        if (p.n_regs > 0) {
            s << "    double ";
            for (int r = 0; r < p.n_regs; ++r) s << (r ? ", " : "") << reg(r) << " = 0.0";
            s << ";\n";
        }
    };
    // the prologue replays the constant-only part of the program (every CONST load and every instruction with a slot)
    o << "__device__ __forceinline__ void knp_prog_" << id << "_pre(const double* __restrict__ C, double* U) {\n";
    if (n_slots > 0) {
        declare(o);
        std::fill(uni.begin(), uni.end(), 0);
        for (int i = 0; i < p.n_instr; ++i) {
            const int op = p.h_code[4 * i], d = p.h_code[4 * i + 1], a = p.h_code[4 * i + 2], b = p.h_code[4 * i + 3];
            if (op != KNP_OP_CONST && slot[(size_t)i] < 0) continue;
            o << "    ";
            if (!emit_statement(o, op, d, a, b)) return false;
            if (slot[(size_t)i] >= 0) o << " U[" << slot[(size_t)i] << "] = " << reg(d) << ";";
            o << "\n";
        }
    }
    o << "}\n\n";
    o << "__device__ __forceinline__ void knp_prog_" << id
      << "(const double* __restrict__ C, const double* U, const double* ki, const double* ke, double phim, const double* aux, const double* xq, double* I) {\n";
    declare(o);
    for (int i = 0; i < p.n_instr; ++i) {
        const int op = p.h_code[4 * i], d = p.h_code[4 * i + 1], a = p.h_code[4 * i + 2], b = p.h_code[4 * i + 3];
        o << "    ";
        if (slot[(size_t)i] >= 0) o << reg(d) << " = U[" << slot[(size_t)i] << "];";
        else if (!emit_statement(o, op, d, a, b)) return false;
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
        << "struct DevParams { double dt, F, C_M, psi; double z[3], Di[3], De[3]; double dz2i[3], dz2e[3], rFz[3], cmFz[3], rF; };\n"
        << "struct FieldPtrs { const double* ki[3]; const double* ke[3]; const double* phim; const double* aux[KNP_MAX_AUX]; };\n"
        << "#define KNP_JIT_UNIFORMS " << KNP_JIT_UNIFORMS << "\n"
        << "__device__ __forceinline__ double powi_d(double x, int e);\n\n";
    int n_aux_used = 0;     // auxiliary nodal fields the programs read (highest slot + 1): the kernel interpolates exactly these
    int xmask = 0;          // coordinate axes the programs read
    for (size_t i = 0; i < progs.size(); ++i) {
        if (progs[i].h_code.size() != (size_t)4 * progs[i].n_instr) { err = "program code not retained"; return false; }
        for (int k = 0; k < progs[i].n_instr; ++k)
            if (progs[i].h_code[4 * k] == KNP_OP_AUX) n_aux_used = std::max(n_aux_used, std::min(progs[i].h_code[4 * k + 2] + 1, (int)KNP_MAX_AUX));
            else if (progs[i].h_code[4 * k] == KNP_OP_X) xmask |= 1 << (progs[i].h_code[4 * k + 2] & 3);
        if (!emit_program(src, (int)i, progs[i])) { err = "unknown opcode"; return false; }
    }
    src << "#define KNP_JIT_NAUX " << n_aux_used << "\n#define KNP_JIT_XMASK " << xmask << "\n\n";
    src << "__device__ __forceinline__ void knp_jit_pre(int prog, const double* __restrict__ C, double* U) {\n    switch (prog) {\n";
    for (size_t i = 0; i < progs.size(); ++i)
        src << "        case " << i << ": knp_prog_" << i << "_pre(C, U); break;\n";
    src << "        default: break;\n    }\n}\n\n";
    src << "__device__ __forceinline__ void knp_jit_eval(int prog, const double* __restrict__ C, const double* U, const double* ki, const double* ke,\n"
           "                                             double phim, const double* aux, const double* xq, double* I) {\n    switch (prog) {\n";
    for (size_t i = 0; i < progs.size(); ++i)
        src << "        case " << i << ": knp_prog_" << i << "(C, U, ki, ke, phim, aux, xq, I); break;\n";
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
    if (const char* dump = getenv("KNP_JIT_DUMP")) {      // developer aid: generated source + code object (llvm-objdump -d, readelf --notes)
        std::ofstream(std::string(dump) + "/knp_gamma_jit.hip") << text;
        std::ofstream(std::string(dump) + "/knp_gamma_jit.hsaco", std::ios::binary).write(code.data(), (std::streamsize)code.size());
    }
    std::lock_guard<std::mutex> lock(cache_mutex());
    cache()[key] = code;
    return true;
}

// Test hook (no device, no context): does this bytecode become a kernel for `arch` ("gfx950")?  0 on success; the
// generated source / compiler log is copied to `log`.
extern "C" int knp_host_thread_count(void) { return knp_host_threads(); }

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
    const int qv = getenv("KNP_GAMMA_QV") ? atoi(getenv("KNP_GAMMA_QV")) : 1;     // points in flight per lane of the 4-lane kernel
    const char* many_name = qv == 9 ? "knp_gamma_vec_3d_many" : "knp_gamma_vec_3d_q1";
    if (hipModuleGetFunction(&f3m, mod, many_name) != hipSuccess) { (void)hipGetLastError(); f3m = nullptr; }
    ctx->jit_module = mod;
    ctx->jit_fn[0] = f2;
    ctx->jit_fn[1] = f3;
    ctx->jit_fn[2] = f3m;
    ctx->jit_msg = "native";
}
