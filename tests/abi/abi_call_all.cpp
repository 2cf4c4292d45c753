// Calls EVERY entry point of include/libzkp_hip.h from C++ through the header alone (no Python, no ctypes): what a Rust / C++
// host binding would do.  Built by __graft_entry__.build() (host compile + link against libzkp_hip.so, no GPU needed), run
// on the GPU box by tests/test_gpu_abi_cpp.py.  Prints "abi_call_all ok: N symbols" and exits 0 when every call behaved.
#include "../../include/libzkp_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <set>
#include <string>
#include <thread>
#include <vector>
#include <sys/resource.h>
#include <hip/hip_runtime_api.h>

static std::set<std::string> called;
static int failures = 0;
#define CALLED(name) called.insert(#name)
#define CHECK(cond)                                                                                   \
    do {                                                                                              \
        if (!(cond)) { std::fprintf(stderr, "FAIL %s:%d: %s   last_error=%s\n", __FILE__, __LINE__, #cond, zkp_hip_last_error()); failures++; } \
    } while (0)

static std::vector<uint8_t> slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static std::vector<uint8_t> seeds_for(size_t n, uint8_t salt) { std::vector<uint8_t> s(32 * n); for (size_t i = 0; i < s.size(); i++) s[i] = (uint8_t)(i * 7 + salt); return s; }

int main(int argc, char** argv) {
    const std::string gold = argc > 1 ? argv[1] : "tests/golden";
    // ---- lifecycle
    CHECK(zkp_hip_init(0) == 0); CALLED(zkp_hip_init);
    CHECK(zkp_hip_device_count() == 1); CALLED(zkp_hip_device_count);
    CHECK(zkp_hip_use_device(0) == 0); CHECK(zkp_hip_use_device(3) == ZKP_HIP_E_ARGUMENT); CALLED(zkp_hip_use_device);
    CHECK(std::strlen(zkp_hip_last_error()) > 0); CALLED(zkp_hip_last_error);
    zkp_hip_set_window_budget(0); CALLED(zkp_hip_set_window_budget);
    zkp_hip_set_subbatches(1); CALLED(zkp_hip_set_subbatches);
    zkp_hip_set_msm_variant(100); CALLED(zkp_hip_set_msm_variant);
    zkp_hip_profile_enable(1); CALLED(zkp_hip_profile_enable);

    // ---- sizes
    CHECK(zkp_hip_range_proof_bytes(64) == ZKP_HIP_RANGE_PROOF_BYTES && zkp_hip_range_proof_bytes(12) == 0); CALLED(zkp_hip_range_proof_bytes);
    CHECK(zkp_hip_threshold_proof_bytes(64) == ZKP_HIP_THRESHOLD_PROOF_BYTES); CALLED(zkp_hip_threshold_proof_bytes);
    CHECK(zkp_hip_consistency_proof_bytes(3) == 10 + 4 + 96 + 2 * (4 + 672 + 32) + 32); CALLED(zkp_hip_consistency_proof_bytes);
    CHECK(zkp_hip_improvement_max_bytes() == 3527); CALLED(zkp_hip_improvement_max_bytes);

    // ---- range: prove (host buffers), verify, device-pointer entry
    const uint64_t n = 5;
    std::vector<uint64_t> v = {0, 7, 50, 99, 100}, lo(n, 0), hi(n, 100);
    auto sd = seeds_for(n, 1);
    std::vector<uint8_t> rp(n * 1478); std::vector<uint32_t> rl(n); std::vector<int32_t> rs(n);
    CHECK(zkp_hip_prove_range_batch(n, v.data(), lo.data(), hi.data(), 64, sd.data(), rp.data(), 1478, rl.data(), rs.data()) == 0); CALLED(zkp_hip_prove_range_batch);
    for (uint64_t i = 0; i < n; i++) CHECK(rl[i] == 1478 && rs[i] == 0 && rp[i * 1478] == 2 && rp[i * 1478 + 1] == 1);
    std::vector<uint8_t> ok(n);
    CHECK(zkp_hip_verify_range_batch(n, rp.data(), 1478, rl.data(), lo.data(), hi.data(), ok.data()) == 0); CALLED(zkp_hip_verify_range_batch);
    for (uint64_t i = 0; i < n; i++) CHECK(ok[i] == 1);
    std::vector<uint64_t> hi_bad(n, 98);
    CHECK(zkp_hip_verify_range_batch(n, rp.data(), 1478, rl.data(), lo.data(), hi_bad.data(), ok.data()) == 0);
    CHECK(ok[0] == 0);                                           // wrong bounds (bulletproofs.rs:704)
    {
        uint64_t *dv, *dlo, *dhi; uint8_t *dsd, *dout; uint32_t* dlen; int32_t* dst;
        CHECK(hipMalloc((void**)&dv, 8 * n) == hipSuccess); CHECK(hipMalloc((void**)&dlo, 8 * n) == hipSuccess); CHECK(hipMalloc((void**)&dhi, 8 * n) == hipSuccess);
        CHECK(hipMalloc((void**)&dsd, 32 * n) == hipSuccess); CHECK(hipMalloc((void**)&dout, 1478 * n) == hipSuccess);
        CHECK(hipMalloc((void**)&dlen, 4 * n) == hipSuccess); CHECK(hipMalloc((void**)&dst, 4 * n) == hipSuccess);
        (void)hipMemcpy(dv, v.data(), 8 * n, hipMemcpyHostToDevice); (void)hipMemcpy(dlo, lo.data(), 8 * n, hipMemcpyHostToDevice);
        (void)hipMemcpy(dhi, hi.data(), 8 * n, hipMemcpyHostToDevice); (void)hipMemcpy(dsd, sd.data(), 32 * n, hipMemcpyHostToDevice);
        (void)hipMemset(dout, 0, 1478 * n);
        int any = -1;
        CHECK(zkp_hip_prove_range_batch_device(n, dv, dlo, dhi, 64, dsd, dout, 1478, dlen, dst, nullptr, &any) == 0 && any == 0); CALLED(zkp_hip_prove_range_batch_device);
        std::vector<uint8_t> back(n * 1478); (void)hipMemcpy(back.data(), dout, back.size(), hipMemcpyDeviceToHost);
        CHECK(back == rp);                                       // same seeds, same bytes as the host-buffer entry
        (void)hipFree(dv); (void)hipFree(dlo); (void)hipFree(dhi); (void)hipFree(dsd); (void)hipFree(dout); (void)hipFree(dlen); (void)hipFree(dst);
    }

    // ---- threshold / consistency
    std::vector<uint64_t> tv = {10, 20, 30, 5}; std::vector<uint32_t> tc = {3, 1}; std::vector<uint64_t> th = {50, 6};
    std::vector<uint8_t> tp(2 * 762); std::vector<uint32_t> tl(2); std::vector<int32_t> ts(2);
    CHECK(zkp_hip_prove_threshold_batch(2, tv.data(), tc.data(), th.data(), 64, sd.data(), tp.data(), 762, tl.data(), ts.data()) == 1); CALLED(zkp_hip_prove_threshold_batch);
    CHECK(ts[0] == 0 && tl[0] == 762 && ts[1] == ZKP_HIP_INVALID_INPUT && tl[1] == 0);      // 5 < 6: sum below the threshold
    CHECK(zkp_hip_verify_threshold_batch(1, tp.data(), 762, tl.data(), th.data(), ok.data()) == 0 && ok[0] == 1); CALLED(zkp_hip_verify_threshold_batch);
    std::vector<uint64_t> cd = {1, 5, 5, 9}; std::vector<uint32_t> cc = {4};
    const uint64_t cstride = zkp_hip_consistency_proof_bytes(4);
    std::vector<uint8_t> cp(cstride); std::vector<uint32_t> cl(1); std::vector<int32_t> cs(1);
    CHECK(zkp_hip_prove_consistency_batch(1, cd.data(), cc.data(), sd.data(), cp.data(), cstride, cl.data(), cs.data()) == 0 && cl[0] == cstride); CALLED(zkp_hip_prove_consistency_batch);
    CHECK(zkp_hip_verify_consistency_batch(1, cp.data(), cstride, cl.data(), ok.data()) == 0 && ok[0] == 1); CALLED(zkp_hip_verify_consistency_batch);

    // ---- Groth16: key generation (small seed), key loading, commitments, equality / membership prove + verify
    {
        std::vector<uint8_t> seed(32, 9); uint64_t pkl = 0, vkl = 0;
        CHECK(zkp_hip_groth16_generate_key(0, seed.data(), nullptr, 0, &pkl, nullptr, 0, &vkl) == 0 && pkl > 100000 && vkl > 0); CALLED(zkp_hip_groth16_generate_key);
    }
    const auto pk_eq = slurp(gold + "/equality_mimc_pk.bin"), pk_mem = slurp(gold + "/membership_mimc_pk.bin");
    CHECK(!pk_eq.empty() && !pk_mem.empty());
    CHECK(zkp_hip_groth16_load_key(0, pk_eq.data(), pk_eq.size()) == 0 && zkp_hip_groth16_load_key(1, pk_mem.data(), pk_mem.size()) == 0); CALLED(zkp_hip_groth16_load_key);
    {   // default key tables: radix 2^13 (the knee), unless the environment opted into something else
        uint32_t wb = 0, un = 9; uint64_t tb = 0;
        CHECK(zkp_hip_groth16_key_info(0, &wb, &un, &tb) == 0 && wb >= 8 && wb <= 15 && un <= 1 && tb > 0); CALLED(zkp_hip_groth16_key_info);
        if (!std::getenv("ZKP_HIP_G16_TABLE_BUDGET_MB") && !std::getenv("ZKP_HIP_G16_WBITS")) CHECK(wb <= 13 && un == 0);
        CHECK(zkp_hip_groth16_key_info(7, &wb, &un, &tb) == ZKP_HIP_E_ARGUMENT);
    }
    std::vector<uint64_t> ev = {42, 43}; std::vector<uint8_t> com(64);
    CHECK(zkp_hip_snark_commit_value_batch(2, ev.data(), com.data()) == 0 && std::memcmp(com.data(), com.data() + 32, 32) != 0); CALLED(zkp_hip_snark_commit_value_batch);
    std::vector<uint8_t> ep(2 * 298); std::vector<uint32_t> el(2); std::vector<int32_t> es(2);
    CHECK(zkp_hip_prove_equality_batch(2, ev.data(), ev.data(), sd.data(), ep.data(), 298, el.data(), es.data()) == 0 && el[0] == 298); CALLED(zkp_hip_prove_equality_batch);
    CHECK(std::memcmp(ep.data() + 266, com.data(), 32) == 0);    // the envelope's commitment is commit_value_snark(value)
    CHECK(zkp_hip_verify_equality_batch(2, ep.data(), 298, el.data(), ok.data()) == 0 && ok[0] == 1 && ok[1] == 1); CALLED(zkp_hip_verify_equality_batch);
    ep[12] ^= 1;                                                 // tests/integration.rs:78-85
    CHECK(zkp_hip_verify_equality_batch(1, ep.data(), 298, el.data(), ok.data()) == 0 && ok[0] == 0);
    std::vector<uint64_t> mv = {25}, ms = {10, 20, 25, 30}; std::vector<uint32_t> mc = {4};
    const uint64_t mstride = 10 + 4 + 32 + 256 + 32;
    std::vector<uint8_t> mp(mstride); std::vector<uint32_t> ml(1); std::vector<int32_t> mst(1);
    CHECK(zkp_hip_prove_membership_batch(1, mv.data(), ms.data(), mc.data(), sd.data(), mp.data(), mstride, ml.data(), mst.data()) == 0 && ml[0] == mstride); CALLED(zkp_hip_prove_membership_batch);
    CHECK(zkp_hip_verify_membership_batch(1, mp.data(), mstride, ml.data(), ok.data()) == 0 && ok[0] == 1); CALLED(zkp_hip_verify_membership_batch);

    // ---- improvement (STARK): host and device entries, verification
    std::vector<uint64_t> io = {30, 5}, in2 = {50, 5};
    std::vector<uint8_t> ip(2 * 3527); std::vector<uint32_t> il(2); std::vector<int32_t> is(2);
    CHECK(zkp_hip_prove_improvement_batch(2, io.data(), in2.data(), ip.data(), 3527, il.data(), is.data()) == 1 && is[0] == 0 && is[1] == ZKP_HIP_INVALID_INPUT); CALLED(zkp_hip_prove_improvement_batch);
    CHECK(zkp_hip_verify_improvement_batch(1, ip.data(), 3527, il.data(), io.data(), ok.data()) == 0 && ok[0] == 1); CALLED(zkp_hip_verify_improvement_batch);
    {
        uint64_t *dold, *dnew; uint8_t* dout; uint32_t* dlen;
        CHECK(hipMalloc((void**)&dold, 8) == hipSuccess); CHECK(hipMalloc((void**)&dnew, 8) == hipSuccess); CHECK(hipMalloc((void**)&dout, 3527) == hipSuccess); CHECK(hipMalloc((void**)&dlen, 4) == hipSuccess);
        (void)hipMemcpy(dold, io.data(), 8, hipMemcpyHostToDevice); (void)hipMemcpy(dnew, in2.data(), 8, hipMemcpyHostToDevice);
        CHECK(zkp_hip_prove_improvement_batch_device(1, dold, dnew, dout, 3527, dlen, nullptr) == 0); CALLED(zkp_hip_prove_improvement_batch_device);
        std::vector<uint8_t> back(il[0]); (void)hipMemcpy(back.data(), dout, il[0], hipMemcpyDeviceToHost);
        CHECK(std::memcmp(back.data(), ip.data(), il[0]) == 0);  // deterministic: both entries give the same envelope
        (void)hipFree(dold); (void)hipFree(dnew); (void)hipFree(dout); (void)hipFree(dlen);
    }

    // ---- mixed batch: capacity query, shard plan, one-shot call, staged call, device results
    std::vector<zkp_hip_op> ops(6); std::vector<uint64_t> lists = {10, 20, 25, 30, 10, 20, 30, 1, 5, 9};
    std::memset(ops.data(), 0, ops.size() * sizeof(zkp_hip_op));
    ops[0].kind = ZKP_HIP_OP_RANGE; ops[0].a = 7; ops[0].b = 0; ops[0].c = 100;
    ops[1].kind = ZKP_HIP_OP_EQUALITY; ops[1].a = 42; ops[1].b = 42;
    ops[2].kind = ZKP_HIP_OP_MEMBERSHIP; ops[2].a = 25; ops[2].count = 4; ops[2].list_off = 0;
    ops[3].kind = ZKP_HIP_OP_IMPROVEMENT; ops[3].a = 3; ops[3].b = 9;
    ops[4].kind = ZKP_HIP_OP_THRESHOLD; ops[4].a = 50; ops[4].count = 3; ops[4].list_off = 4;
    ops[5].kind = ZKP_HIP_OP_CONSISTENCY; ops[5].count = 3; ops[5].list_off = 7;
    auto bs = seeds_for(6, 3);
    uint64_t cap = 0;
    CHECK(zkp_hip_process_batch_bytes(6, ops.data(), &cap) == 0 && cap == 1478 + 298 + (10 + 4 + 32 + 256 + 32) + 3527 + 762 + zkp_hip_consistency_proof_bytes(3)); CALLED(zkp_hip_process_batch_bytes);
    std::vector<uint32_t> owner(6);
    CHECK(zkp_hip_plan_shards(6, ops.data(), 2, owner.data()) == 0); CALLED(zkp_hip_plan_shards);
    std::vector<uint8_t> out(cap); std::vector<uint64_t> off(7); std::vector<int32_t> st(6);
    CHECK(zkp_hip_process_batch(6, ops.data(), lists.data(), bs.data(), out.data(), cap, off.data(), st.data()) == 0); CALLED(zkp_hip_process_batch);
    const uint8_t scheme[6] = {1, 2, 4, 5, 3, 6};
    for (int i = 0; i < 6; i++) CHECK(st[i] == 0 && off[i + 1] > off[i] && out[off[i]] == 2 && out[off[i] + 1] == scheme[i]);
    zkp_hip_batch* B = nullptr;
    CHECK(zkp_hip_batch_stage(6, ops.data(), lists.data(), bs.data(), &B) == 0 && B != nullptr); CALLED(zkp_hip_batch_stage);
    CHECK(zkp_hip_batch_max_bytes(B) == cap); CALLED(zkp_hip_batch_max_bytes);
    CHECK(zkp_hip_batch_prove(B) == 0); CALLED(zkp_hip_batch_prove);
    {   // two batches in flight on the shard's two lanes: same bytes as the blocking call
        zkp_hip_batch* B2 = nullptr;
        CHECK(zkp_hip_batch_stage(6, ops.data(), lists.data(), bs.data(), &B2) == 0);
        CHECK(zkp_hip_batch_prove_async(B) == 0 && zkp_hip_batch_prove_async(B2) == 0); CALLED(zkp_hip_batch_prove_async);
        CHECK(zkp_hip_batch_wait(B2) == 0 && zkp_hip_batch_wait(B) == 0); CALLED(zkp_hip_batch_wait);
        std::vector<uint8_t> o2(cap); std::vector<uint64_t> f2(7); std::vector<int32_t> s2(6);
        CHECK(zkp_hip_batch_fetch(B2, o2.data(), cap, f2.data(), s2.data()) == 0 && f2 == off && std::memcmp(o2.data(), out.data(), off[6]) == 0);
        zkp_hip_batch_free(B2);
    }
    std::vector<uint8_t> out2(cap); std::vector<uint64_t> off2(7); std::vector<int32_t> st2(6);
    CHECK(zkp_hip_batch_fetch(B, out2.data(), cap, off2.data(), st2.data()) == 0 && off2 == off && std::memcmp(out.data(), out2.data(), off[6]) == 0); CALLED(zkp_hip_batch_fetch);
    {
        uint8_t* dres; uint64_t* doff; uint64_t nops = 0;
        CHECK(hipMalloc((void**)&dres, cap) == hipSuccess); CHECK(hipMalloc((void**)&doff, 8 * 7) == hipSuccess);
        CHECK(zkp_hip_batch_device_results(B, 0, dres, cap, doff, &nops, nullptr) == 0 && nops == 6); CALLED(zkp_hip_batch_device_results);
        std::vector<uint8_t> back(off[6]); std::vector<uint64_t> boff(7);
        (void)hipMemcpy(back.data(), dres, off[6], hipMemcpyDeviceToHost); (void)hipMemcpy(boff.data(), doff, 56, hipMemcpyDeviceToHost);
        CHECK(boff == off && std::memcmp(back.data(), out.data(), off[6]) == 0);
        (void)hipFree(dres); (void)hipFree(doff);
    }
    zkp_hip_batch_free(B); CALLED(zkp_hip_batch_free);

    // ---- the exception barrier (SURVEY 8b "never abort"; batch.rs:126-130): absurd sizes are argument errors, an allocation failure
    // inside the library is a negative code with a message -- never an exception through this C++/C frame -- and the library still
    // proves the same bytes afterwards
    {
        zkp_hip_batch* BX = nullptr;
        CHECK(zkp_hip_batch_stage(1ull << 40, ops.data(), lists.data(), bs.data(), &BX) == ZKP_HIP_E_ARGUMENT && BX == nullptr && std::strstr(zkp_hip_last_error(), "batch too large"));
        CHECK(zkp_hip_process_batch(1ull << 62, ops.data(), lists.data(), bs.data(), out.data(), cap, off.data(), st.data()) == ZKP_HIP_E_ARGUMENT);
        CHECK(zkp_hip_prove_range_batch(1ull << 33, v.data(), lo.data(), hi.data(), 64, sd.data(), rp.data(), 1478, rl.data(), rs.data()) == ZKP_HIP_E_ARGUMENT);
        CHECK(zkp_hip_verify_equality_batch(1ull << 35, ep.data(), 298, el.data(), ok.data()) == ZKP_HIP_E_ARGUMENT);
        // counts[] that sum past 2^32: rejected before a single list value is read (the lists here hold ten values)
        std::vector<uint32_t> huge = {0xffffffffu, 0xffffffffu}; std::vector<uint64_t> th2 = {1, 1};
        std::vector<uint8_t> tp2(2 * 762); std::vector<uint32_t> tl2(2); std::vector<int32_t> ts2(2);
        CHECK(zkp_hip_prove_threshold_batch(2, lists.data(), huge.data(), th2.data(), 64, sd.data(), tp2.data(), 762, tl2.data(), ts2.data()) == ZKP_HIP_E_ARGUMENT && std::strstr(zkp_hip_last_error(), "value lists too long"));
        CHECK(zkp_hip_prove_consistency_batch(2, lists.data(), huge.data(), sd.data(), tp2.data(), 762, tl2.data(), ts2.data()) == ZKP_HIP_E_ARGUMENT);
        std::vector<zkp_hip_op> bad = ops; bad[4].count = 0xffffffffu; bad[5].count = 0xffffffffu;
        CHECK(zkp_hip_batch_stage(6, bad.data(), lists.data(), bs.data(), &BX) == ZKP_HIP_E_ARGUMENT && BX == nullptr);
        // address space capped just above what the process holds now: staging 2^21 ops (host vectors of tens of MB, a pinned image of
        // ~150 MB) cannot be allocated.  The call reports it; then the limit is lifted and the small batch proves bit-exactly again.
        const uint64_t big = 1ull << 21;
        std::vector<zkp_hip_op> many(big);
        for (uint64_t i = 0; i < big; i++) { many[i] = ops[0]; many[i].a = i & 63; }
        uint64_t vm_kb = 0;
        { std::ifstream f("/proc/self/status"); std::string w; while (f >> w) if (w == "VmSize:") { f >> vm_kb; break; } }
        struct rlimit old_lim, lim;
        CHECK(vm_kb > 0 && getrlimit(RLIMIT_AS, &old_lim) == 0);
        lim = old_lim; lim.rlim_cur = (vm_kb << 10) + (24ull << 20);
        CHECK(setrlimit(RLIMIT_AS, &lim) == 0);
        const int rc_oom = zkp_hip_batch_stage(big, many.data(), lists.data(), nullptr, &BX);
        const std::string msg_oom = zkp_hip_last_error();
        CHECK(setrlimit(RLIMIT_AS, &old_lim) == 0);
        CHECK(rc_oom < 0 && BX == nullptr && !msg_oom.empty());
        std::fprintf(stderr, "staging 2^21 ops under RLIMIT_AS: rc %d, \"%s\"\n", rc_oom, msg_oom.c_str());
        std::vector<uint8_t> o4(cap); std::vector<uint64_t> f4(7); std::vector<int32_t> s4(6);
        CHECK(zkp_hip_process_batch(6, ops.data(), lists.data(), bs.data(), o4.data(), cap, f4.data(), s4.data()) == 0 && f4 == off && std::memcmp(o4.data(), out.data(), off[6]) == 0);
    }

    // ---- profiling counters saw the MSM launches of the calls above
    double kms = 0; uint64_t launches = 0, adds = 0;
    CHECK(zkp_hip_profile_read(&kms, &launches, &adds, 0) == 0 && launches > 0 && adds > 0); CALLED(zkp_hip_profile_read);
    CHECK(zkp_hip_profile_read_kernel(ZKP_HIP_KERNEL_MSM_BN254_G1, &kms, &launches, &adds, 1) == 0 && launches > 0); CALLED(zkp_hip_profile_read_kernel);

    // ---- two shards on this one GPU (the same HIP device twice), then back to nothing
    zkp_hip_shutdown(); CALLED(zkp_hip_shutdown);
    const int devs[2] = {0, 0};
    CHECK(zkp_hip_init_devices(2, devs) == 0 && zkp_hip_device_count() == 2); CALLED(zkp_hip_init_devices);
    CHECK(zkp_hip_groth16_load_key(0, pk_eq.data(), pk_eq.size()) == 0 && zkp_hip_groth16_load_key(1, pk_mem.data(), pk_mem.size()) == 0);
    std::vector<uint8_t> out3(cap); std::vector<uint64_t> off3(7); std::vector<int32_t> st3(6);
    CHECK(zkp_hip_process_batch(6, ops.data(), lists.data(), bs.data(), out3.data(), cap, off3.data(), st3.data()) == 0);
    CHECK(off3 == off && std::memcmp(out3.data(), out.data(), off[6]) == 0);   // sharded == unsharded, byte for byte
    {   // two host threads, each staging / proving / fetching its own batches on the two shards at the same time (the reference calls
        // process_batch from rayon workers, batch.rs:125-130): every shard's worker takes the two callers' jobs in turn, nobody's job is
        // lost or answered by the other's completion -- every result equals the single-threaded bytes
        int bad_runs[2] = {0, 0};
        auto caller = [&](int t) {
            for (int it = 0; it < 6; it++) {
                zkp_hip_batch* Bt = nullptr;
                std::vector<uint8_t> ot(cap); std::vector<uint64_t> ft(7); std::vector<int32_t> stt(6);
                if (zkp_hip_batch_stage(6, ops.data(), lists.data(), bs.data(), &Bt) != 0 || zkp_hip_batch_prove_async(Bt) != 0 || zkp_hip_batch_wait(Bt) != 0 ||
                    zkp_hip_batch_fetch(Bt, ot.data(), cap, ft.data(), stt.data()) != 0 || ft != off || std::memcmp(ot.data(), out.data(), off[6]) != 0) bad_runs[t]++;
                zkp_hip_batch_free(Bt);
            }
        };
        std::thread ta(caller, 0), tb(caller, 1);
        ta.join(); tb.join();
        CHECK(bad_runs[0] == 0 && bad_runs[1] == 0);
    }
    zkp_hip_shutdown();
    CHECK(zkp_hip_device_count() == 0);

    const size_t expected = 45;                                  // declarations in include/libzkp_hip.h (tests/test_abi.py counts them too)
    if (called.size() != expected) { std::fprintf(stderr, "called %zu of %zu entry points\n", called.size(), expected); failures++; }
    if (failures) { std::fprintf(stderr, "abi_call_all: %d failure(s)\n", failures); return 1; }
    std::printf("abi_call_all ok: %zu symbols\n", called.size());
    return 0;
}
