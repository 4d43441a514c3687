// librangeldm_hip runtime: model objects, weight packing, per-batch execution plans (one activation arena + an
// ordered list of kernel launches), HIP-graph captured sampling loop, and the C ABI of include/rangeldm_hip.h.
//
// Network structure follows diffusers UNet2DModel.forward [3P; SURVEY.md A.2] as configured by
// ldm/train_unconditional.py:237-242 and the sgm Encoder/Decoder forward
// (vae/sgm/modules/diffusionmodules/model.py:852-896,1024-1057); the loop follows ldm/pipelines.py:353-367.
#include "../../include/rangeldm_hip.h"
#include "kernels.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <vector>

namespace rldm {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }

// ---------------------------------------------------------------------------------------------------------------
// device memory helpers
// ---------------------------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { reset(); }
    void reset() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    int alloc(size_t n) {
        reset();
        if (n == 0) n = 16;
        RLDM_HIP_CHECK(hipMalloc(&p, n));
        bytes = n;
        return 0;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

static int upload(DevBuf& b, const void* host, size_t bytes) {
    if (b.alloc(bytes)) return 1;
    RLDM_HIP_CHECK(hipMemcpy(b.p, host, bytes, hipMemcpyHostToDevice));
    return 0;
}

// first-fit offset allocator over one arena; plans are built twice (dry run for the peak, then for real)
struct Arena {
    struct Blk { size_t off, size; };
    std::vector<Blk> free_list;
    size_t top = 0, peak = 0;
    size_t alloc(size_t n) {
        n = (n + 255) & ~(size_t)255;
        for (size_t i = 0; i < free_list.size(); ++i) {
            if (free_list[i].size >= n) {
                size_t off = free_list[i].off;
                free_list[i].off += n;
                free_list[i].size -= n;
                if (free_list[i].size == 0) free_list.erase(free_list.begin() + i);
                return off;
            }
        }
        size_t off = top;
        top += n;
        peak = std::max(peak, top);
        return off;
    }
    void release(size_t off, size_t n) {
        n = (n + 255) & ~(size_t)255;
        free_list.push_back({off, n});
        std::sort(free_list.begin(), free_list.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
        for (size_t i = 0; i + 1 < free_list.size();) {
            if (free_list[i].off + free_list[i].size == free_list[i + 1].off) {
                free_list[i].size += free_list[i + 1].size;
                free_list.erase(free_list.begin() + i + 1);
            } else {
                ++i;
            }
        }
        if (!free_list.empty() && free_list.back().off + free_list.back().size == top) {
            top = free_list.back().off;
            free_list.pop_back();
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// layers: host fp32 parameters + lazily packed device images
// ---------------------------------------------------------------------------------------------------------------
struct NormParams {
    DevBuf gamma, beta;
    int C = 0;
};

struct ConvLayer {
    std::string name;
    int Cout = 0, Cin = 0, ksize = 3;          // real sizes
    std::vector<float> w, b;                   // host fp32, (Cout, Cin, k, k) / (Cout)
    // residual phase fused into this conv's K loop (conv_igemm.hip): a 1x1 map over R channels of the block input.
    // sc_w = conv_shortcut weights (Cout, R) [its bias is folded into b], or empty + sc_identity for a plain `x + h`.
    int R = 0;
    bool sc_identity = false;
    std::vector<float> sc_w;
    struct Packed {
        DevBuf w, bias;
        int ntile_n = 0, Cin_pad = 0;
    };
    std::map<std::pair<int, int>, std::unique_ptr<Packed>> packed;   // (BN, CK) -> image

    // [ntile_n][Cin_pad/CK * taps + R/CK][BN][CK + 8] bf16; channel rows >= Cout and channels >= Cin are zero
    int get_packed(int BN, int CK, int Cin_pad, Packed** out) {
        auto key = std::make_pair(BN, CK);
        auto it = packed.find(key);
        if (it != packed.end()) {
            RLDM_REQUIRE(it->second->Cin_pad == Cin_pad, "conv layer reused with a different channel padding");
            *out = it->second.get();
            return 0;
        }
        RLDM_REQUIRE(R % CK == 0, "conv " + name + ": residual channels not a multiple of the channel chunk");
        const int taps = ksize * ksize;
        const int ntile = (Cout + BN - 1) / BN;
        const int ncc = Cin_pad / CK, ncb = R / CK;
        const int RSE = CK + 8;
        const size_t stages = (size_t)ncc * taps + ncb;
        std::vector<bf16_t> img((size_t)ntile * stages * BN * RSE, 0);
        for (int n = 0; n < Cout; ++n) {
            const int nt = n / BN, nr = n % BN;
            for (int c = 0; c < Cin; ++c) {
                const int cc = c / CK, ck = c % CK;
                for (int tap = 0; tap < taps; ++tap) {
                    const float v = w[((size_t)n * Cin + c) * taps + tap];
                    img[(((size_t)nt * stages + (size_t)cc * taps + tap) * BN + nr) * RSE + ck] = f32_to_bf16(v);
                }
            }
            for (int c = 0; c < R; ++c) {
                const float v = sc_identity ? (c == n ? 1.f : 0.f) : sc_w[(size_t)n * R + c];
                img[(((size_t)nt * stages + (size_t)ncc * taps + c / CK) * BN + nr) * RSE + c % CK] = f32_to_bf16(v);
            }
        }
        std::vector<float> bias((size_t)ntile * BN, 0.f);
        for (int n = 0; n < Cout; ++n) bias[n] = b[n];
        auto pk = std::make_unique<Packed>();
        if (upload(pk->w, img.data(), img.size() * sizeof(bf16_t))) return 1;
        if (upload(pk->bias, bias.data(), bias.size() * sizeof(float))) return 1;
        pk->ntile_n = ntile;
        pk->Cin_pad = Cin_pad;
        *out = pk.get();
        packed[key] = std::move(pk);
        return 0;
    }

    // conv_stream.hip image: MFMA A-fragment order, one contiguous stream of 1 KiB k-steps per (32-channel tile, k-group):
    // [Cout/32][KG][Cin_pad/64 chunks x 9 taps x 4/KG k-steps, then R/64 chunks x 4/KG k-steps][64 lanes][8 bf16] + 32 KiB
    // of zeros (the ring's read-ahead past the last stream: up to 18 fragments of 1 KiB in the 64-pixel instance); k-group kg owns the k-steps [kg*4/KG, (kg+1)*4/KG) of every
    // tap of a chunk; lane l holds channel 32*t + (l & 31), k = 8*(l >> 5) .. +8 of the step
    int get_streampacked(int Cin_pad, int KG, Packed** out) {
        auto key = std::make_pair(-2, KG);
        auto it = packed.find(key);
        if (it != packed.end()) {
            RLDM_REQUIRE(it->second->Cin_pad == Cin_pad, "conv layer reused with a different channel padding");
            *out = it->second.get();
            return 0;
        }
        RLDM_REQUIRE(ksize == 3 && (Cout % 32 == 0 || Cout < 32) && Cin_pad % 64 == 0 && R % 64 == 0, "conv " + name + ": not stream-packable");
        const int SPT = 4 / KG, NCC = Cin_pad / 64, NCB = R / 64, nsteps = (NCC * 9 + NCB) * SPT;
        const int ntile32 = (Cout + 31) / 32;        // (fewer than 32 output channels -- conv_regw.hip's conv_out: one tile, zero rows and zero bias behind them)
        std::vector<bf16_t> img((size_t)ntile32 * KG * nsteps * 512 + 16384, 0);
        auto at = [&](int n, int ks, int step, int k) -> bf16_t& {      // ks: 16-channel group within the 64-channel chunk
            const size_t stream = (size_t)(n / 32) * KG + ks / SPT;
            return img[((stream * nsteps + step + ks % SPT) * 64 + (k / 8) * 32 + n % 32) * 8 + k % 8];
        };
        for (int n = 0; n < Cout; ++n) {
            for (int c = 0; c < Cin; ++c)
                for (int tap = 0; tap < 9; ++tap)
                    at(n, (c % 64) / 16, ((c / 64) * 9 + tap) * SPT, c % 16) = f32_to_bf16(w[((size_t)n * Cin + c) * 9 + tap]);
            for (int c = 0; c < R; ++c) {
                const float v = sc_identity ? (c == n ? 1.f : 0.f) : sc_w[(size_t)n * R + c];
                at(n, (c % 64) / 16, (NCC * 9 + c / 64) * SPT, c % 16) = f32_to_bf16(v);
            }
        }
        auto pk = std::make_unique<Packed>();
        if (upload(pk->w, img.data(), img.size() * sizeof(bf16_t))) return 1;
        std::vector<float> bias((size_t)ntile32 * 32, 0.f);
        for (int n = 0; n < Cout; ++n) bias[n] = b[n];
        if (upload(pk->bias, bias.data(), bias.size() * sizeof(float))) return 1;
        pk->ntile_n = 0;
        pk->Cin_pad = Cin_pad;
        *out = pk.get();
        packed[key] = std::move(pk);
        return 0;
    }

    // conv_regw.hip, conv_c16_kernel (the input layer: <= 16 input channels): [Cout / 32][9 taps][64 lanes][8 bf16] -- lane l of a fragment holds
    // channel 32 t + (l & 31), input channels 8 (l >> 5) .. + 8
    int get_c16packed(Packed** out) {
        auto key = std::make_pair(-4, 1);
        auto it = packed.find(key);
        if (it != packed.end()) {
            *out = it->second.get();
            return 0;
        }
        RLDM_REQUIRE(ksize == 3 && Cout % 32 == 0 && Cin <= 16 && R == 0, "conv " + name + ": not a 16-channel input layer");
        std::vector<bf16_t> img((size_t)(Cout / 32) * 9 * 512, 0);
        for (int n = 0; n < Cout; ++n)
            for (int c = 0; c < Cin; ++c)
                for (int tap = 0; tap < 9; ++tap)
                    img[(((size_t)(n / 32) * 9 + tap) * 64 + (c / 8) * 32 + n % 32) * 8 + c % 8] = f32_to_bf16(w[((size_t)n * Cin + c) * 9 + tap]);
        auto pk = std::make_unique<Packed>();
        if (upload(pk->w, img.data(), img.size() * sizeof(bf16_t))) return 1;
        if (upload(pk->bias, b.data(), b.size() * sizeof(float))) return 1;
        pk->ntile_n = 0;
        pk->Cin_pad = 16;
        *out = pk.get();
        packed[key] = std::move(pk);
        return 0;
    }

    // conv_stream.hip, sub-pixel form of nearest x2 + 3x3 (conv_stream_body.h, SUB): per (32-channel tile, parity pw * 2 + ph) one stream
    // [Cin_pad/64 chunks][2 x 2 taps (w-major)][4 k-steps][64 lanes][8 bf16] of SUMMED weights -- along each axis parity 0 reads inputs
    // (x - 1, x) through (k[0], k[1] + k[2]), parity 1 reads (x, x + 1) through (k[0] + k[1], k[2]); summed in fp32, rounded once
    int get_subpixpacked(int Cin_pad, Packed** out) {
        auto key = std::make_pair(-3, 1);
        auto it = packed.find(key);
        if (it != packed.end()) {
            RLDM_REQUIRE(it->second->Cin_pad == Cin_pad, "conv layer reused with a different channel padding");
            *out = it->second.get();
            return 0;
        }
        RLDM_REQUIRE(ksize == 3 && Cout % 32 == 0 && Cin_pad % 64 == 0 && R == 0, "conv " + name + ": not sub-pixel-packable");
        const int NCC = Cin_pad / 64, nsteps = NCC * 16;
        std::vector<bf16_t> img((size_t)(Cout / 32) * 4 * nsteps * 512 + 16384, 0);
        static const int lo[2][2] = {{0, 1}, {0, 2}}, hi[2][2] = {{0, 2}, {1, 2}};    // [parity][tap]: original taps lo .. hi summed
        for (int n = 0; n < Cout; ++n)
            for (int c = 0; c < Cin; ++c)
                for (int pw = 0; pw < 2; ++pw)
                    for (int ph = 0; ph < 2; ++ph)
                        for (int i = 0; i < 2; ++i)
                            for (int j = 0; j < 2; ++j) {
                                float v = 0.f;
                                for (int a = lo[pw][i]; a <= hi[pw][i]; ++a)
                                    for (int bb = lo[ph][j]; bb <= hi[ph][j]; ++bb) v += w[((size_t)n * Cin + c) * 9 + a * 3 + bb];
                                const size_t stream = (size_t)(n / 32) * 4 + pw * 2 + ph;
                                const int step = ((c / 64) * 4 + i * 2 + j) * 4 + (c % 64) / 16, k = c % 16;
                                img[((stream * nsteps + step) * 64 + (k / 8) * 32 + n % 32) * 8 + k % 8] = f32_to_bf16(v);
                            }
        auto pk = std::make_unique<Packed>();
        if (upload(pk->w, img.data(), img.size() * sizeof(bf16_t))) return 1;
        if (upload(pk->bias, b.data(), b.size() * sizeof(float))) return 1;
        pk->ntile_n = 0;
        pk->Cin_pad = Cin_pad;
        *out = pk.get();
        packed[key] = std::move(pk);
        return 0;
    }

    // conv_small.hip image: MFMA A-fragment order, one contiguous stream of 1 KiB k-steps per (32-channel tile, k-group):
    // [Cout/32][KG][taps*CPT main steps (tap-major) + RPT residual steps][64 lanes][8 bf16] + one zero fragment; k-group kg
    // owns the 16-channel groups kg, kg + KG, ... of every tap; lane l of a step holds channel 32*t + (l & 31),
    // k = 8*(l >> 5) .. +8.  `no_res`: the (identity) residual is added by the kernel's epilogue, not multiplied here.
    int get_fragpacked(int Cin_pad, int KG, bool no_res, Packed** out) {
        auto key = std::make_pair(no_res ? -1 : 0, KG);
        auto it = packed.find(key);
        if (it != packed.end()) {
            RLDM_REQUIRE(it->second->Cin_pad == Cin_pad, "conv layer reused with a different channel padding");
            *out = it->second.get();
            return 0;
        }
        const int taps = ksize * ksize, Rp = no_res ? 0 : R;
        RLDM_REQUIRE(Cout % 32 == 0 && Cin_pad % (16 * KG) == 0 && Rp % (16 * KG) == 0, "conv " + name + ": not fragment-packable");
        const int CPT = Cin_pad / 16 / KG, RPT = Rp / 16 / KG, nmine = taps * CPT + RPT;
        std::vector<bf16_t> img((size_t)(Cout / 32) * KG * nmine * 512 + 512, 0);
        auto at = [&](int n, int step, int c16, int k) -> bf16_t& {     // step within the stream of k-group c16 % KG
            const size_t stream = (size_t)(n / 32) * KG + c16 % KG;
            return img[((stream * nmine + step) * 64 + (k / 8) * 32 + n % 32) * 8 + k % 8];
        };
        for (int n = 0; n < Cout; ++n) {
            for (int c = 0; c < Cin; ++c)
                for (int tap = 0; tap < taps; ++tap)
                    at(n, tap * CPT + (c / 16) / KG, c / 16, c % 16) = f32_to_bf16(w[((size_t)n * Cin + c) * taps + tap]);
            for (int c = 0; c < Rp; ++c) {
                const float v = sc_identity ? (c == n ? 1.f : 0.f) : sc_w[(size_t)n * R + c];
                at(n, taps * CPT + (c / 16) / KG, c / 16, c % 16) = f32_to_bf16(v);
            }
        }
        auto pk = std::make_unique<Packed>();
        if (upload(pk->w, img.data(), img.size() * sizeof(bf16_t))) return 1;
        if (upload(pk->bias, b.data(), b.size() * sizeof(float))) return 1;
        pk->ntile_n = 0;
        pk->Cin_pad = Cin_pad;
        *out = pk.get();
        packed[key] = std::move(pk);
        return 0;
    }

    // (round 5) conv_small.hip's 16-channel tiles: one stream of 1 KiB k-steps per (16-channel tile, k-group of 8):
    // [Cout/16][8][taps*CPT main steps (tap-major) + RPT residual steps][64 lanes][8 bf16] + one zero fragment; a k-step is 32 input
    // channels (v_mfma_f32_16x16x32_bf16: lane l holds channel 16*t + (l & 15), k = 8*(l >> 4) .. +8); k-group kg owns the 32-channel
    // groups kg, kg + 8, ... of every tap.
    int get_fragpacked16(int Cin_pad, bool no_res, Packed** out) {
        auto key = std::make_pair(no_res ? -6 : -5, 16);
        auto it = packed.find(key);
        if (it != packed.end()) {
            RLDM_REQUIRE(it->second->Cin_pad == Cin_pad, "conv layer reused with a different channel padding");
            *out = it->second.get();
            return 0;
        }
        const int taps = ksize * ksize, Rp = no_res ? 0 : R, KG = 8;
        RLDM_REQUIRE(Cout % 16 == 0 && Cin_pad % (32 * KG) == 0 && Rp % (32 * KG) == 0, "conv " + name + ": not 16-channel fragment-packable");
        const int CPT = Cin_pad / 32 / KG, RPT = Rp / 32 / KG, nmine = taps * CPT + RPT;
        std::vector<bf16_t> img((size_t)(Cout / 16) * KG * nmine * 512 + 512, 0);
        auto at = [&](int n, int step, int c32, int k) -> bf16_t& {     // step within the stream of k-group c32 % KG
            const size_t stream = (size_t)(n / 16) * KG + c32 % KG;
            return img[((stream * nmine + step) * 64 + (k / 8) * 16 + n % 16) * 8 + k % 8];
        };
        for (int n = 0; n < Cout; ++n) {
            for (int c = 0; c < Cin; ++c)
                for (int tap = 0; tap < taps; ++tap)
                    at(n, tap * CPT + (c / 32) / KG, c / 32, c % 32) = f32_to_bf16(w[((size_t)n * Cin + c) * taps + tap]);
            for (int c = 0; c < Rp; ++c) {
                const float v = sc_identity ? (c == n ? 1.f : 0.f) : sc_w[(size_t)n * R + c];
                at(n, taps * CPT + (c / 32) / KG, c / 32, c % 32) = f32_to_bf16(v);
            }
        }
        auto pk = std::make_unique<Packed>();
        if (upload(pk->w, img.data(), img.size() * sizeof(bf16_t))) return 1;
        if (upload(pk->bias, b.data(), b.size() * sizeof(float))) return 1;
        pk->ntile_n = 0;
        pk->Cin_pad = Cin_pad;
        *out = pk.get();
        packed[key] = std::move(pk);
        return 0;
    }
};

// (ConvLayer::get_streampacked is defined with the struct above)
struct ParamStore {
    std::map<std::string, std::vector<float>> host;
    std::map<std::string, int64_t> expected;    // name -> numel
    bool finalized = false;

    int set(const char* name, const float* data, int64_t numel) {
        auto it = expected.find(name);
        RLDM_REQUIRE(it != expected.end(), std::string("unexpected parameter key: ") + name);
        RLDM_REQUIRE(it->second == numel, std::string("size mismatch for ") + name + ": expected " +
                                              std::to_string(it->second) + " got " + std::to_string(numel));
        host[name].assign(data, data + numel);
        finalized = false;
        return 0;
    }
    int check_complete() const {
        for (auto& kv : expected)
            RLDM_REQUIRE(host.count(kv.first), std::string("missing parameter key: ") + kv.first);
        return 0;
    }
    void expect_conv(const std::string& n, int co, int ci, int k) {
        expected[n + ".weight"] = (int64_t)co * ci * k * k;
        expected[n + ".bias"] = co;
    }
    void expect_lin(const std::string& n, int co, int ci) {
        expected[n + ".weight"] = (int64_t)co * ci;
        expected[n + ".bias"] = co;
    }
    void expect_norm(const std::string& n, int c) {
        expected[n + ".weight"] = c;
        expected[n + ".bias"] = c;
    }
    void expect_resnet(const std::string& p, int ci, int co, int temb) {
        expect_norm(p + ".norm1", ci);
        expect_conv(p + ".conv1", co, ci, 3);
        if (temb) expect_lin(p + ".time_emb_proj", co, temb);
        expect_norm(p + ".norm2", co);
        expect_conv(p + ".conv2", co, co, 3);
        if (ci != co) expect_conv(p + ".conv_shortcut", co, ci, 1);
    }
    void expect_attn(const std::string& p, int c) {
        expect_norm(p + ".group_norm", c);
        expect_lin(p + ".to_q", c, c);
        expect_lin(p + ".to_k", c, c);
        expect_lin(p + ".to_v", c, c);
        expect_lin(p + ".to_out.0", c, c);
    }
};

// ---------------------------------------------------------------------------------------------------------------
// execution plan
// ---------------------------------------------------------------------------------------------------------------
struct Tensor {
    size_t off = 0;
    int B = 0, W = 0, H = 0, C = 0;
    int id = -1;
    size_t st_off = 0;     // per-channel partial statistics [B][P][C] float2 (P == 0: none)
    int P = 0;
    int prod = -1;         // ordinal of the Builder::conv call that produced it (-1: something else)
    size_t st_bytes() const { return (size_t)B * P * C * sizeof(float2); }
    size_t bytes() const { return (size_t)B * W * H * C * sizeof(bf16_t); }
    bool valid() const { return id >= 0; }
};

struct PlanIO {
    // conv_in sources (fp32 NCHW, device)
    const float* sample = nullptr;
    int sample_channels = 0;
    float sample_scale = 1.f;
    int pos_encoding = 0;
    const float* cond = nullptr;
    int cond_channels = 0;
    // output (fp32 NCHW, device)
    float* out = nullptr;
    // time-embedding table
    const float* temb = nullptr;
    const int* step_ptr = nullptr;
    int temb_rows_per_step = 1;
    int temb_per_sample = 0;
    // sampler only: the step index advanced by the step's first launch, the scheduler step in conv_out's epilogue
    int* step_inc = nullptr;
    SchedFuse sch = {};
    // sampler only: conv_out's epilogue also writes the next step's conv_in input (sch.pack = xin), so the plan's pack_input launch is
    // skipped and conv_in advances the step index; the sampler packs x_T itself once per call
    bool pack_fused = false;
    bf16_t* xin = nullptr;         // the plan's conv_in input tensor [B][W][H][xin_ld] (never recycled inside the plan)
    int xin_ld = 0;
};

struct Op {
    std::function<int(hipStream_t)> fn;
    std::string name;       // kernel (template instance) the op launches
    double flops = 0;       // algorithmic FLOPs (2*MACs) of this launch
    double bytes = 0;       // algorithmic HBM bytes: every input / weight / output touched once
    std::function<bool()> active;   // optional: false -> the op launches nothing this run (the sampler's fused pack_input) and is skipped
    std::string tag;                // layer name(s), for RLDM_PRINT_PLAN
};

struct KernelStat {
    int launches = 0;
    double ms = 0, flops = 0, bytes = 0;
};

struct Plan {
    int B = 0, W = 0, H = 0;
    int flags = 0;                 // routing options of THIS plan (the bits of rldm_debug_set_flags), in force while it is built
    DevBuf arena;
    DevBuf tickets;                // split-K arrival counters of every conv in the plan
    std::vector<std::unique_ptr<DevBuf>> trunk_bufs;   // phase records / cluster counters of the persistent trunk launches
    DevBuf trunk_error;            // device int: a trunk launch gave up a wait (1) or found a cluster off its XCD (2)
    PlanIO io;
    std::vector<Op> ops;
    double flops = 0;
    int run(hipStream_t s) {
        for (auto& o : ops)
            if ((!o.active || o.active()) && o.fn(s)) return 1;
        return 0;
    }
    // same, with a timestamp launch around every op (rldm_debug_graph_trace: the timeline of the ops as they run
    // back to back inside the captured graph, which a host-side profiler cannot see without spacing them out)
    int run_stamped(hipStream_t s, unsigned long long* slots, int cap) {
        if (launch_stamp(slots, s)) return 1;
        int n = 0;                              // (ops that launch nothing this run leave no stamp)
        for (size_t i = 0; i < ops.size(); ++i) {
            if (ops[i].active && !ops[i].active()) continue;
            if (ops[i].fn(s)) return 1;
            ++n;
            if (n < cap && launch_stamp(slots + n, s)) return 1;
        }
        return 0;
    }
    // eager pass with a HIP event pair around every launch on `s` (the stream the kernels run on)
    int run_profiled(hipStream_t s, std::map<std::string, KernelStat>& stats) {
        std::vector<hipEvent_t> ev(ops.size() + 1);
        for (auto& e : ev) RLDM_HIP_CHECK(hipEventCreate(&e));
        RLDM_HIP_CHECK(hipEventRecord(ev[0], s));
        std::vector<char> ran(ops.size(), 1);
        for (size_t i = 0; i < ops.size(); ++i) {
            ran[i] = !ops[i].active || ops[i].active();
            if (ran[i] && ops[i].fn(s)) return 1;
            RLDM_HIP_CHECK(hipEventRecord(ev[i + 1], s));
        }
        RLDM_HIP_CHECK(hipStreamSynchronize(s));
        for (size_t i = 0; i < ops.size(); ++i) {
            if (!ran[i]) continue;
            float ms = 0.f;
            RLDM_HIP_CHECK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            KernelStat& k = stats[ops[i].name];
            k.launches += 1;
            k.ms += ms;
            k.flops += ops[i].flops;
            k.bytes += ops[i].bytes;
        }
        for (auto& e : ev) (void)hipEventDestroy(e);
        return 0;
    }
};

struct Layers {
    std::map<std::string, std::unique_ptr<ConvLayer>> conv;
    std::map<std::string, std::unique_ptr<NormParams>> norm;
    ConvLayer* get_conv(const std::string& n) {
        auto it = conv.find(n);
        return it == conv.end() ? nullptr : it->second.get();
    }
    NormParams* get_norm(const std::string& n) {
        auto it = norm.find(n);
        return it == norm.end() ? nullptr : it->second.get();
    }
    int add_conv(ParamStore& ps, const std::string& n, int co, int ci, int k) {
        auto L = std::make_unique<ConvLayer>();
        L->name = n;
        L->Cout = co;
        L->Cin = ci;
        L->ksize = k;
        L->w = ps.host.at(n + ".weight");
        L->b = ps.host.at(n + ".bias");
        conv[n] = std::move(L);
        return 0;
    }
    int add_norm(ParamStore& ps, const std::string& n, int c) {
        auto N = std::make_unique<NormParams>();
        N->C = c;
        if (upload(N->gamma, ps.host.at(n + ".weight").data(), c * sizeof(float))) return 1;
        if (upload(N->beta, ps.host.at(n + ".bias").data(), c * sizeof(float))) return 1;
        norm[n] = std::move(N);
        return 0;
    }
    // fused q|k|v projection; q rows pre-scaled by log2(e)/sqrt(head_dim) so attention works in exp2 units
    int add_qkv(ParamStore& ps, const std::string& p, int c, int head_dim) {
        auto L = std::make_unique<ConvLayer>();
        L->name = p + ".qkv";
        L->Cout = 3 * c;
        L->Cin = c;
        L->ksize = 1;
        const float qs = 1.4426950408889634f / std::sqrt((float)head_dim);
        const char* names[3] = {".to_q", ".to_k", ".to_v"};
        for (int i = 0; i < 3; ++i) {
            const auto& w = ps.host.at(p + names[i] + ".weight");
            const auto& b = ps.host.at(p + names[i] + ".bias");
            const float s = i == 0 ? qs : 1.f;
            for (float v : w) L->w.push_back(v * s);
            for (float v : b) L->b.push_back(v * s);
        }
        conv[L->name] = std::move(L);
        return 0;
    }
};

// Per-head MFMA A fragments of the merged q|k|v Linear for attention_qkv_d8_kernel: w [3C][C] (q rows already scaled by
// log2(e)/sqrt(8)), b [3C] -> img [heads][C/16][64 lanes][8] bf16 (row = lane & 31: 0-7 q, 8-15 k, 16-23 v, 24-31 zero;
// channels 16*ks + 8*(lane >> 5) ..+8), bias [heads][32].
static void pack_attn_head_frags(const float* w, const float* b, int C, std::vector<bf16_t>& img, std::vector<float>& bias) {
    const int heads = C / 8, nks = C / 16;
    img.assign((size_t)heads * nks * 512, 0);
    bias.assign((size_t)heads * 32, 0.f);
    for (int h = 0; h < heads; ++h)
        for (int row = 0; row < 24; ++row) {
            const int src = (row / 8) * C + h * 8 + row % 8;            // q | k | v rows of the merged Linear
            bias[(size_t)h * 32 + row] = b[src];
            for (int c = 0; c < C; ++c)
                img[(((size_t)h * nks + c / 16) * 64 + ((c % 16) / 8) * 32 + row) * 8 + c % 8] = f32_to_bf16(w[(size_t)src * C + c]);
        }
}

struct ConvArgs {
    ConvLayer* layer = nullptr;
    Tensor x0, x1;                 // x1 optional concat
    int stride = 1, pad_mode = 0, up = 1;
    NormParams* gn = nullptr;      // GroupNorm prologue over cat[x0, x1] (needs their statistics)
    float eps = 1e-5f;
    int groups = 32;
    int silu = 0;
    int temb_off = -1;             // channel offset into the temb table row
    Tensor r0, r1;                 // residual-phase sources (layer->R channels in total), at output resolution
    bool want_stats = false;       // emit per-channel statistics of the output (a GroupNorm will read it)
    bool out_f32_nchw = false;     // conv_out: write plan->io.out
    bool own_image = false;        // conv_small route with ONE tile per image (the producer normalises for its consumers)
    bool first_of_step = false;    // conv_in: advances the sampler's step index when the plan's pack_input launch is fused away
};

// Producer-side GroupNorm: which convs write a normalised (+ activated) copy of their output for which
// consuming GroupNorm.  Filled by a recording pass over the network walk (the walk is the same in every pass, so a conv is
// identified by its ordinal), read by the sizing and the real pass.
struct ViewPlan {
    struct Part { int prod; int ch_off; };
    struct Cons {
        std::vector<Part> parts;
        int Ctot = 0, silu = 0, groups = 32;
        float eps = 1e-5f;
        bool ok = false;               // the consumer can read a pre-activated tensor (conv_small 3x3, fused attention)
        bool use = false;              // decided: ok and every part's producer can emit
    };
    struct Emit { const NormParams* norm; int ch_off; int Ctot; int silu; int groups; float eps; };
    std::vector<char> can_emit;        // by conv ordinal
    std::vector<int> out_c;            // output channels, by conv ordinal (tuning aid below)
    std::map<const NormParams*, Cons> cons;
    std::map<int, std::vector<Emit>> emit;
    void decide() {
        std::map<int, int> nper;
        for (auto& kv : cons) {
            Cons& c = kv.second;
            bool good = c.ok && c.groups > 0 && c.Ctot % c.groups == 0;
            const int cpg = good ? c.Ctot / c.groups : 0;
            good = good && cpg >= 1 && cpg <= 32 && (cpg & (cpg - 1)) == 0;
            for (auto& pt : c.parts)
                good = good && pt.prod >= 0 && pt.prod < (int)can_emit.size() && can_emit[pt.prod] && pt.ch_off % cpg == 0 &&
                       nper[pt.prod] < 3;
            c.use = good;
            if (!good) continue;
            for (auto& pt : c.parts) {
                emit[pt.prod].push_back({kv.first, pt.ch_off, c.Ctot, c.silu, c.groups, c.eps});
                nper[pt.prod]++;
            }
        }
    }
};

// routing / ablation switches (rldm_debug_set_flags); RLDM_DBG_FLAGS seeds them for A/B runs of unmodified drivers
static int g_dbg_flags = getenv("RLDM_DBG_FLAGS") ? atoi(getenv("RLDM_DBG_FLAGS")) : 0;
// ... and the options of the plan being built (Plan::flags: rldm_sampler_config::plan_flags, rldm_unet_set_plan_flags, the fall-back
// of a sampler whose persistent launches failed their self-check) -- scoped to that plan, unlike the process-wide word above
static thread_local int t_plan_flags = 0;
static inline int dbg() { return g_dbg_flags | t_plan_flags; }
// second word of process-wide tuning switches (rldm_debug_set_flags2 / RLDM_DBG_FLAGS2), round 4:
//   1 full-resolution conv_stream launches keep the 8-wave 256-pixel workgroups (default: 4-wave 128 x 128 workgroups, two per CU)
//   2 the 128x8 level keeps the 8-wave 128 x 64 x 4-k-group workgroups (default: 4-wave 128 x 64 x 2 k-groups, two per CU)
//   4 the VAE's 64-channel level keeps the 8-wave 256 x 64 instance
// 128 stride-2 convs stay on the generic kernel (default: the 64-pixel conv_stream tile where its grid fits)
//  16 the 128x8 level keeps the 128 x 64 x 4-k-group tiles (default: 64 pixels x 128 channels x 2 k-groups)
//   8 the 4-wave full-resolution convs stay launches of their own (default: phases of trunk variant 4, two workgroups per CU)
//   bits 8..15: conv_stream experiment switches (ConvParams::exp); bits 16..23: (n + 1) = trunk variant 4's start offset n
static int g_dbg_flags2 = getenv("RLDM_DBG_FLAGS2") ? atoi(getenv("RLDM_DBG_FLAGS2")) : 0;
static inline int dbg2() { return g_dbg_flags2; }
static const int kInst4MinBlocks = getenv("RLDM_INST4_MIN") ? atoi(getenv("RLDM_INST4_MIN")) : 96;    // (tuning; 192 -> 96: nuScenes at 4 images 87.4 -> 89.4,
                                                                                           //  KITTI at 8 images 134.4 -> 137.5 img/s)
static constexpr int kTrunkSkewDefault = 0;     // (trunk variant 4: start offset of the second image group, x 1024 cycles; measured 0 / 4 / 8 / 16 /
                                                //  24 -> 230.6 / 230.1 / 230.8 / 228.7 / 223.7 img/s; RLDM_DBG_FLAGS2 = (n + 1) << 16: n)
static unsigned long long* g_ts_buf = nullptr;   // rldm_debug_timestamps: device [4][64] s_memtime stamps
static int g_force_bm = 0, g_force_bn = 0, g_force_ks = 0;
static int g_split_auto = 0;     // automatic split-K is off: the in-launch combine costs more than the idle CUs     // rldm_debug_force_tile: tuning override (0: automatic)

static constexpr bool kStreamPairs64Default = false;
static thread_local int g_concurrent_plans = 1;   // plans being built will share the device with this many of their kind (sampler chains)

struct TileChoice {
    ConvTile tile;
    int TW = 1, TH = 1, ksplit = 1;
};

// output-pixel tile of a block: as tall as the image allows (<= 16 beams), then as wide as BM allows
static void pixel_tile(int BM, int Wout, int Hout, int stride, int* TW, int* TH) {
    int th = 1;
    while (th * 2 <= Hout && th * 2 <= 16 && Hout % (th * 2) == 0) th *= 2;
    int tw = 1;
    while (tw * 2 * th <= BM && Wout % (tw * 2) == 0) tw *= 2;
    (void)stride;
    *TW = tw;
    *TH = th;
}

static TileChoice choose_tile(long long B, int Wout, int Hout, int stride, int N, int Cin_pad, int C0, int R, int R0,
                              int taps, bool nchw, bool gn) {
    TileChoice c;
    const int KW = taps == 9 ? 3 : 1;
    // a channel chunk never straddles a concat boundary
    auto ck_ok = [&](int ck) { return Cin_pad % ck == 0 && C0 % ck == 0 && R % ck == 0 && R0 % ck == 0; };
    // instance for a (BM, BN): the widest channel chunk it exists with
    auto pick = [&](int BM, int BN, ConvTile* u, int* tw, int* th) {
        const int cks[3] = {64, 32, 16};
        for (int ck : cks) {
            if (!ck_ok(ck)) continue;
            ConvTile v;
            v.BM = BM; v.BN = BN; v.CK = ck; v.taps = taps;
            if (!conv_tile_supported(v)) continue;
            pixel_tile(BM, Wout, Hout, stride, tw, th);
            // shrink the tile until the halo fits the instance's register staging capacity
            bool ok = true;
            while (((*tw - 1) * stride + KW) * ((*th - 1) * stride + KW) > conv_max_halo_slots(v)) {
                if (*tw > 1) *tw >>= 1; else if (*th > 1) *th >>= 1; else { ok = false; break; }
            }
            if (!ok) continue;
            *u = v;
            return true;
        }
        return false;
    };
    auto blocks = [&](const ConvTile& u, int tw, int th) {
        return B * (Wout / tw) * (Hout / th) * ((N + u.BN - 1) / u.BN);
    };
    const int bn_pref = N <= 32 ? 32 : (N <= 64 ? 64 : 128);
    // candidates in order of preference (preferred channel tile first, largest pixel tile first); take the first that
    // fills the chip, else the one with the most blocks.  Pass 1 also accepts tiles that are mostly padding; pass 2 any
    // channel tile (small test configurations have few instances).
    const int bms[3] = {256, 128, 64};
    std::vector<int> bns = {bn_pref};
    if (bn_pref > 64) bns.push_back(64);
    long long best_blocks = -1;
    TileChoice best;
    bool found = false;
    for (int pass = 0; pass < 3 && best_blocks < 0; ++pass) {
        if (pass == 2) bns = {32, 64, 128};
        for (int bi = 0; bi < 3 && !found; ++bi)
            for (size_t ni = 0; ni < bns.size() && !found; ++ni) {
                ConvTile u;
                int tw, th;
                if (!pick(bms[bi], bns[ni], &u, &tw, &th)) continue;
                // more than half empty: try a smaller BM first -- except under stride 2, where the halo capacity shrinks the
                // pixel tile of every instance and the 256-pixel one-tap pipeline still wins when it fills the chip (measured)
                if (pass == 0 && tw * th * 2 <= u.BM && bi < 2 && !(stride == 2 && bi == 0)) continue;
                const long long nb = blocks(u, tw, th);
                if (nb > best_blocks) {
                    best_blocks = nb;
                    best.tile = u;
                    best.TW = tw;
                    best.TH = th;
                }
                // enough blocks?  A 1x1 conv with a GroupNorm prologue (attention q/k/v) re-normalises its input once per
                // channel tile, so a wide tile on half the CUs beats two rounds of narrow ones (measured, tools/bench_conv.py)
                if (nb >= ((taps == 1 && gn) ? 128 : 200)) found = true;
            }
    }
    if (g_force_bm && g_force_bn) {
        ConvTile u;
        int tw, th;
        if (pick(g_force_bm, g_force_bn, &u, &tw, &th)) {
            best.tile = u;
            best.TW = tw;
            best.TH = th;
            best_blocks = blocks(u, tw, th);
        }
    }
    c = best;
    if (best_blocks < 0) {
        c.tile.BM = 64; c.tile.BN = 64; c.tile.CK = 16; c.tile.taps = taps;
        return c;                    // caller reports "no kernel instance" if this one does not exist either
    }
    // split-K over channel chunks when the tile grid leaves most CUs idle (fp32 NCHW outputs skip it: tiny N anyway)
    c.ksplit = 1;
    const int ncc = Cin_pad / c.tile.CK;
    if (!nchw) {
        int ks = 1;
        if (g_split_auto)
            while (best_blocks * ks * 2 <= 320 && ks * 2 <= ncc && ncc % (ks * 2) == 0) ks *= 2;
        c.ksplit = ks;
        if (g_force_ks > 0 && g_force_ks <= std::max(1, ncc)) c.ksplit = g_force_ks;
    }
    return c;
}

struct Builder {
    Plan* plan;
    bool dry;
    Arena arena;
    int next_id = 0;
    std::map<int, int> refs;
    std::map<int, Tensor> live;
    char* base = nullptr;
    int launches = 0;
    int temb_ld = 0;               // row stride of the time-embedding table (0: network has none)
    int* ticket_ptr = nullptr;     // split-K arrival counters (plan->tickets, zeroed at allocation; null in the dry pass)
    int tickets = 0;
    ViewPlan* vp = nullptr;        // producer-side GroupNorm plan (null: off)
    bool recording = false;        // the pass that fills *vp (a dry pass)
    int conv_ord = 0;
    int groups_hint = 32;
    std::map<const NormParams*, Tensor> view_tensors;      // consumer norm -> its pre-activated input (alive until consumed)
    const std::vector<ViewPlan::Emit>* cur_emit = nullptr; // of the conv being built

    // recording pass: GroupNorm `n` over cat[x0, x1] is consumed by something that could read a pre-activated tensor instead
    void record_consumer(const NormParams* n, const Tensor& x0, const Tensor& x1, int silu, bool ok, int groups, float eps) {
        if (!vp || !recording || !n) return;
        ViewPlan::Cons c;
        c.parts.push_back({x0.prod, 0});
        if (x1.valid()) c.parts.push_back({x1.prod, x0.C});
        c.Ctot = x0.C + (x1.valid() ? x1.C : 0);
        c.silu = silu;
        c.groups = groups;
        c.eps = eps;
        c.ok = ok && x0.W * x0.H == (x1.valid() ? x1.W * x1.H : x0.W * x0.H);
        vp->cons[n] = c;
    }
    // sizing / real pass: the pre-activated input of the consumer that owns norm `n`, if the plan has one
    bool take_view(const NormParams* n, Tensor* v) {
        if (!vp || recording || !n) return false;
        auto it = vp->cons.find(n);
        if (it == vp->cons.end() || !it->second.use) return false;
        auto vt = view_tensors.find(n);
        if (vt == view_tensors.end()) return false;
        *v = vt->second;
        view_tensors.erase(vt);
        return true;
    }

    Tensor make(int B, int W, int H, int C) {
        Tensor t;
        t.B = B; t.W = W; t.H = H; t.C = C;
        t.id = next_id++;
        t.off = arena.alloc(t.bytes());
        refs[t.id] = 1;
        live[t.id] = t;
        return t;
    }
    void add_stats(Tensor& t, int P) {
        t.P = P;
        t.st_off = arena.alloc(t.st_bytes());
        live[t.id] = t;
    }
    void retain(const Tensor& t) { refs[t.id]++; }
    void release(const Tensor& t) {
        if (!t.valid()) return;
        if (trunk_open) {          // (the memory may not be recycled inside the segment)
            deferred.push_back(t);
            return;
        }
        if (--refs[t.id] == 0) {
            arena.release(t.off, t.bytes());
            if (t.P) arena.release(t.st_off, t.st_bytes());
            live.erase(t.id);
        }
    }
    // ---- persistent trunk (trunk.hip): consecutive image-owning conv_small launches collected into one launch ----------------
    struct PendingTrunk {
        std::vector<TrunkPhase> phases;
        std::vector<Op> standalone;   // the same layers as launches of their own (a one-phase segment runs as that: a phase costs its
                                      // record fetch, the padded grid and the arrive for nothing)
        size_t lds = 0;
        int B = 0, ranks = 0;
        int ntile_n = 0, nwn = 1;      // channel tiles per image, 32-channel tiles per workgroup (multi-tile clusters: nwn == 2)
        int variant = 0;               // TrunkParams::variant
        double flops = 0, bytes = 0;
    } pend;
    bool trunk_open = false;
    std::map<int, int> trunk_seen;  // persistent launches built so far, by phase count (RLDM_TS_TRUNK_FIRST)
    std::vector<Tensor> deferred;  // releases held back while a segment is open (clusters of different images drift apart)
    // rldm_debug_set_flags: 1 << 24 keeps every phase a launch of its own with the tiles unchanged (tests: identical results);
    // 1 << 25 also gives the convs back their default tiles (A/B runs of the whole feature)
    static bool trunk_enabled() { return !(dbg() & ((1 << 24) | (1 << 25))); }
    static bool trunk_tiles() { return !(dbg() & (1 << 25)); }
    void note_launch() {           // every launch that is not a trunk phase closes the open segment
        flush_trunk();
        ++launches;
    }
    int flush_trunk() {
        if (!trunk_open) return 0;
        trunk_open = false;
        int rc = 0;
        // the runtime's own answer to "are all these workgroups resident at once?" (variant 4: two per CU) -- asked here, at plan build,
        // not left to the launch's bounded-wait self-check.  A query that FAILS (< 0) changes nothing; a clear "no" keeps the layers launches.
        const int need_per_cu = pend.variant == 4 ? 2 : 1;
        const int resident = (!dry && pend.phases.size() > 1 && !(dbg() & (1 << 29))) ? trunk_max_resident(pend.variant, pend.lds) : need_per_cu;
        const bool not_resident = resident >= 0 && resident < need_per_cu && pend.standalone.size() == pend.phases.size();
        if (not_resident) {
            static bool said = false;
            if (!said) fprintf(stderr, "librangeldm_hip: trunk_kernel<%d> with %zu bytes of LDS: %d workgroup(s) per CU resident, %d needed; "
                               "these layers run as launches of their own\n", pend.variant, pend.lds, resident, need_per_cu);
            said = true;
            launches += (int)pend.standalone.size() - 1;
            for (auto& op : pend.standalone) plan->ops.push_back(op);
        } else if (!dry && pend.phases.size() == 1 && pend.standalone.size() == 1 && !(dbg() & (1 << 29))) {
            plan->ops.push_back(pend.standalone[0]);
        } else if (!dry && !pend.phases.empty()) {
            auto recs = std::make_unique<DevBuf>();
            auto ctrs = std::make_unique<DevBuf>();
            if (upload(*recs, pend.phases.data(), pend.phases.size() * sizeof(TrunkPhase))) return 1;
            if (ctrs->alloc((size_t)pend.B * 32 * 4)) return 1;
            RLDM_HIP_CHECK(hipMemset(ctrs->p, 0, ctrs->bytes));
            if (!plan->trunk_error.p) {
                if (plan->trunk_error.alloc(64)) return 1;
                RLDM_HIP_CHECK(hipMemset(plan->trunk_error.p, 0, 64));
            }
            TrunkParams tp;
            memset(&tp, 0, sizeof(tp));
            tp.phases = recs->as<TrunkPhase>();
            tp.nphases = (int)pend.phases.size();
            tp.B = pend.B;
            tp.ranks = pend.ranks;
            tp.ntile_n = pend.nwn == 1 ? pend.ranks : pend.ntile_n;
            tp.nwn = pend.nwn;
            tp.variant = pend.variant;
            // (variant 4's start offset between the two image groups, x 1024 cycles; RLDM_DBG_FLAGS2 bits 16..23 override the default)
            tp.skew = pend.variant == 4 ? (((dbg2() >> 16) & 255) ? ((dbg2() >> 16) & 255) - 1 : kTrunkSkewDefault) : 0;
            tp.counters = ctrs->as<unsigned>();
            tp.error = plan->trunk_error.as<int>();
            tp.temb_ld = temb_ld;
            plan->trunk_bufs.push_back(std::move(recs));
            plan->trunk_bufs.push_back(std::move(ctrs));
            Plan* pl = plan;
            const size_t lds = pend.lds;
            const bool ts_ok = !getenv("RLDM_TS_TRUNK_FIRST") || ++trunk_seen[(int)pend.phases.size()] == 1;   // (timeline of the FIRST such launch)
            plan->ops.push_back({[tp, lds, pl, ts_ok](hipStream_t st) mutable {
                tp.temb = pl->io.temb;
                tp.step_ptr = pl->io.step_ptr;
                tp.temb_rows_per_step = pl->io.temb_rows_per_step;
                tp.temb_per_sample = pl->io.temb_per_sample;
                tp.ts = (ts_ok && getenv("RLDM_TS_TRUNK") && tp.nphases == atoi(getenv("RLDM_TS_TRUNK"))) ? g_ts_buf : nullptr;
                return launch_trunk(tp, lds, st);
            }, std::string("trunk_kernel<") + (pend.variant == 0 ? "conv_small image tiles" : pend.variant == 1 ? "conv_small 64x64 clusters" :
                                               pend.variant == 2 ? "conv_stream 256x128" : pend.variant == 3 ? "conv_stream 128x64" : pend.variant == 5 ? "conv_stream 64x128" : "conv_stream 128x128 x2/CU") +
                   ", " + std::to_string(pend.phases.size()) + " phases>", pend.flops, pend.bytes});
        }
        pend = PendingTrunk();
        std::vector<Tensor> d;
        d.swap(deferred);
        for (const Tensor& t : d) release(t);
        return rc;
    }

    // attention core as a trunk phase: x arrives normalised, C / 8 heads split over N / 32 = C / 32 workgroups of 8 waves
    static size_t trunk_attention_lds(int L, int C, int HG) { return attention_qkv2_lds_bytes(L, C, HG, 8); }
    bool trunk_attention_ok(const Tensor& x, bool pre) const {
        const int L = x.W * x.H, ranks = x.C / own_tile_channels(x.B, L, x.C);
        // (x.C % 64: attention_qkv2_body loads x in coalesced 64-channel groups and projects in blocks of 4 k-steps, no tail)
        if (!trunk_enabled() || !pre || x.C % 64 != 0 || ranks < 2 || ranks > 16) return false;
        const int HG = (x.C / 8) / ranks, wph = (L + 31) / 32;
        // 8 waves: one query tile per wave of a head (32-token images -- the lowest nuScenes level -- leave every second wave idle:
        // without the phase that level was five persistent launches with an attention launch between each pair)
        if (HG * wph > 8 || 8 % HG != 0 || (8 / HG) < wph || L > 64) return false;
        if (!trunk_grid_fits(ranks, x.B)) return false;
        return trunk_attention_lds(L, x.C, HG) <= 160 * 1024;
    }
    // ... of a multi-tile cluster: raw x + statistics (the fold runs inside the phase), two query tiles per wave; 0: not eligible
    int cluster_attention_ranks(const Tensor& x, bool pre) const {
        const int L = x.W * x.H, ranks = cluster_ranks(x.B, x.C, L);
        if (!cluster_enabled() || (dbg() & 512) || pre || ranks == 0 || x.P <= 0 || (x.C / 8) % ranks != 0 || L % 32 != 0) return 0;
        const int HG = (x.C / 8) / ranks, wph = L / 32;
        if (HG * wph != 16 || x.C > 512) return 0;                // 8 waves x two query tiles
        return trunk_attention_lds(L, x.C, HG) <= 160 * 1024 ? ranks : 0;
    }
    static int device_cus() {
        static int cus = [] { int dev = 0, n = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0; return n; }();
        return std::min(256, cus > 0 ? cus : 256);
    }
    // a persistent launch spin-waits on its own workgroups: ALL of them must be resident at once, whatever else the plan's owner runs
    // beside it -- a sampler's other chains (g_concurrent_plans: two 256-workgroup launches could each hold part of the chip and wait
    // for the rest) and the device's real CU count (partitions / smaller parts)
    // (per_cu = 2: the 4-wave conv_stream variant, whose workgroups are built for two per CU)
    // (round 5) channels per image-owning tile: 16 for the 64-pixel images of the KITTI network's 32x2 level (256 output channels -> 16
    // workgroups per image), when that grid is resident at once; 32 otherwise.  RLDM_OWN16=0 keeps 32 everywhere (A/B runs).
    static bool own16_enabled() { static const bool on = !(getenv("RLDM_OWN16") && atoi(getenv("RLDM_OWN16")) == 0); return on; }
    static int own_tile_channels(int B, int npix, int N) {
        return (own16_enabled() && trunk_tiles() && npix == 64 && N == 256 && trunk_grid_fits(16, B)) ? 16 : 32;    // (trunk_tiles(), not trunk_enabled():
        // the per-layer fall-back -- 1 << 24 -- keeps the persistent launches' tiles, so its results stay identical)
    }
    static bool trunk_grid_fits(int ranks, int B, int per_cu = 1) {
        return 8 * ranks * ((B + 7) / 8) * std::max(1, g_concurrent_plans) <= device_cus() * per_cu;
    }
    void trunk_begin(int B, int ranks, int ntile_n = 0, int nwn = 1, int variant = -1) {
        if (nwn == 1) ntile_n = ranks;
        if (variant < 0) variant = nwn == 1 ? 0 : 1;
        if (trunk_open && (pend.ranks != ranks || pend.B != B || pend.nwn != nwn || pend.ntile_n != ntile_n || pend.variant != variant))
            flush_trunk();
        if (!trunk_open) {
            ++launches;
            trunk_open = true;
            pend.B = B;
            pend.ranks = ranks;
            pend.ntile_n = ntile_n;
            pend.nwn = nwn;
            pend.variant = variant;
        }
    }
    // multi-tile clusters (trunk.hip, kinds 8..13): an image = (N / 64) channel tiles x (pixels / 64) pixel tiles of conv_small's
    // 64 x 64 instance, all of them resident at once and on one XCD (8 * ranks * ceil(B / 8) workgroups <= the chip's CUs).
    // rldm_debug_set_flags(1 << 26) keeps these levels as separate launches (A/B runs)
    // (and only for plans that run ALONE on the device: a 256-workgroup launch of one sampler chain and one of another could each
    //  hold part of the chip and wait for the rest -- sampler_build_plans sets g_concurrent_plans to its number of chains)
    static bool cluster_enabled() { return trunk_enabled() && !(dbg() & (1 << 26)) && g_concurrent_plans <= 1; }
    static int cluster_ranks(int B, int C, int npix) {
        if (C % 64 != 0 || npix % 64 != 0) return 0;
        const int r = (C / 64) * (npix / 64);
        return (r >= 2 && r <= 16 && npix > 64 && trunk_grid_fits(r, B)) ? r : 0;
    }
    void trunk_push_attention(const AttnQkvParams& ap, double fl, double by) {
        TrunkPhase ph;
        memset(&ph, 0, sizeof(ph));
        auto put64 = [&](int at, const void* ptr) {
            const unsigned long long u = (unsigned long long)(uintptr_t)ptr;
            ph.w[at] = (unsigned)u;
            ph.w[at + 1] = (unsigned)(u >> 32);
        };
        put64(TW_X0, ap.x); put64(TW_WPK, ap.wfrag); put64(TW_BIAS, ap.bias); put64(TW_Y, ap.out);
        ph.w[TW_WIN] = ap.L; ph.w[TW_N] = ap.C;
        ph.w[TW_KIND] = ap.st ? TK_ATTN_FOLD : TK_ATTN; ph.w[TW_G] = 0; ph.w[TW_NMINE] = 0; ph.w[TW_TEMBOFF] = (unsigned)-1;
        ph.w[TW_WBYTES] = (unsigned)(3 * ap.C * ap.C * 2);          // q / k / v projection fragments (attention_body.h)
        if (ap.st) {                    // the consumer-side GroupNorm of x (multi-tile clusters)
            put64(TW_ST0, ap.st); put64(TW_GAMMA, ap.gamma); put64(TW_BETA, ap.beta);
            ph.w[TW_P0] = ap.P; ph.w[TW_GROUPS] = ap.groups; ph.w[TW_MAGIC_CPG] = ap.magic_cpg;
            memcpy(&ph.w[TW_INVN], &ap.inv_n, 4); memcpy(&ph.w[TW_EPS], &ap.eps, 4);
        }
        pend.phases.push_back(ph);
        pend.lds = std::max(pend.lds, trunk_attention_lds(ap.L, ap.C, (ap.C / 8) / pend.ranks));
        pend.flops += fl;
        pend.bytes += by;
    }

    template <class T> T* ptr(size_t off) const { return reinterpret_cast<T*>(base + off); }
    bf16_t* tptr(const Tensor& t) const { return t.valid() ? ptr<bf16_t>(t.off) : nullptr; }
    const float2* sptr(const Tensor& t) const { return (t.valid() && t.P) ? ptr<float2>(t.st_off) : nullptr; }

    // per-channel statistics of a tensor that did not come out of a conv epilogue (external inputs in the test entry points)
    int gn_stats(Tensor& x) {
        const int npix = x.W * x.H;
        const int P = std::max(1, std::min(16, npix / 64));
        add_stats(x, P);
        note_launch();
        if (!dry) {
            GnStatsParams g;
            g.x = tptr(x);
            g.C = x.C;
            g.B = x.B;
            g.npix = npix;
            g.P = P;
            g.part = ptr<float2>(x.st_off);
            const double by = (double)x.B * npix * x.C * 2.0;
            plan->ops.push_back({[g](hipStream_t s) { return launch_gn_stats(g, s); }, "gn_stats_kernel", 0.0, by});
        }
        return 0;
    }

    // conv_small.hip route: 3x3 / stride 1 convs over <= 256-pixel images (the 64x4 and 32x2 UNet levels) and every
    // pointwise conv with 128..512 input channels (attention q/k/v and output projections)
    static void small_tile(int bm, int Wout, int Hout, int* tw, int* th) {
        int h = 1;
        while (h * 2 <= Hout && h * 2 <= 8 && Hout % (h * 2) == 0) h *= 2;
        *th = h;
        *tw = bm / h;
    }
    // geometry + channel counts of the route; `epi_res`: the identity residual is added in the epilogue instead of the K loop
    static bool small_params(const ConvArgs& a, int Cin_t, int R_t, int taps, int Wout, int Hout, ConvParams* q, bool* epi_res) {
        if (dbg() & 256) return false;
        if ((taps != 9 && taps != 1) || a.stride != 1 || (a.up != 1 && !(a.up == 2 && taps == 9 && !a.gn && R_t == 0)) || a.out_f32_nchw)
            return false;
        // (128-pixel tiles for the 128x8 level are built and tested but lose to the generic kernel there: the separate
        //  GroupNorm pass over a 12 MB tensor costs more than the faster K loop wins; rldm_debug_set_flags(1024) routes them)
        // small images (C2: the 64x4 / 32x2 levels at batch 16), or few pixels in the whole batch (C1 / C3: 128x8 at batch 1,
        // 128x4 at batch 4): the same 64-pixel tiles, more of them per image
        const bool few_px = (long long)a.x0.B * Wout * Hout <= 4096 && !(dbg() & 65536);
        if (taps == 9 && (a.pad_mode != 0 || (Wout * Hout > ((dbg() & 1024) ? 1024 : 256) && !few_px))) return false;
        // pixel tile: 64; 128 for the 3x3 convs of the 128x8 level (each weight fragment then feeds 4 MFMAs)
        // (32 for 32x1 images: the lowest nuScenes level, which otherwise runs as 8 workgroups of the generic kernel; and for
        //  32x2 images, as two tiles each: twice the workgroups, half the staging / epilogue per workgroup -- level-3 convs
        //  13.4-14.2 -> 13.0 us, +0.5-1 % end to end; rldm_debug_set_flags(524288) keeps the 64-pixel tile: A/B runs, tests)
        int bm = (taps == 9 && Wout * Hout > 256 && !few_px) ? 128 : ((Wout * Hout == 32 || (Wout * Hout == 64 && Hout == 2 && !(dbg() & 524288))) && a.up == 1 ? 32 : 64);
        if (a.own_image) {                      // one tile per image (<= 64 pixels), or not this route
            if (Wout * Hout > 64 || a.up != 1) return false;
            bm = Wout * Hout;
        }
        if (taps == 1 && a.x1.valid()) return false;
        if (!a.gn && a.x1.valid()) return false;
        if (a.layer->Cout % 32 != 0 || (g_force_bm && g_force_bm != 64)) return false;
        int tw, th;
        small_tile(bm, Wout, Hout, &tw, &th);
        if (Wout % tw != 0 || Hout % th != 0 || Wout < 2) return false;
        *epi_res = a.layer->sc_identity && a.r0.valid() && !a.r1.valid() && a.r0.C == a.layer->Cout;
        memset(q, 0, sizeof(*q));
        {
            const bool cat = taps == 9 && a.x1.valid() && small_gn_fused(a, taps);     // a concatenated input normalised by the conv's own staging
            q->C0 = cat ? a.x0.C : Cin_t;       // (else one input tensor: single, or pre-activated by gn_apply)
            q->C1 = cat ? Cin_t - a.x0.C : 0;
        }
        q->R0 = *epi_res ? 0 : (a.r0.valid() ? a.r0.C : 0);
        q->R1 = *epi_res ? 0 : R_t - q->R0;
        q->B = a.x0.B; q->Win = a.x0.W; q->Hin = a.x0.H;
        q->up = a.up; q->stride = 1; q->pad_lo = taps == 9 ? 1 : 0;
        q->Wout = Wout; q->Hout = Hout;
        q->TW = tw; q->TH = th;
        q->colb = conv_small_col_bytes(Cin_t, th, taps);
        q->tiles_h = Hout / th;
        q->tiles_img = (Wout / tw) * q->tiles_h;
        while ((1 << q->th_shift) < th) ++q->th_shift;
        const int cpg = std::max(1, Cin_t / a.groups);
        q->magic_cpg = ((1 << 20) + cpg - 1) / cpg;
        q->gn_inv_n = (float)(1.0 / ((double)a.x0.W * a.x0.H * cpg));
        q->N = a.layer->Cout;
        q->silu = a.silu;
        q->gn_eps = a.eps;
        q->gn_groups = a.groups;
        q->ksplit = 1;
        return true;
    }
    // channel tile; 0: no instance / not worth it.
    // 3x3: 64 channels when that fills the chip, else 32 (twice the blocks, half the weight stream per block).
    // 1x1: the work per block is a few MFMAs and the launch is one latency chain per block, so the route is taken only if
    // the grid fits one round of 256 workgroups -- with the narrowest tile that does (most blocks); else the generic kernel.
    static int small_bn(const ConvParams& q, int taps, bool gn_fused, bool own = false) {
        ConvParams t = q;
        if (gn_fused) t.st0 = reinterpret_cast<const float2*>(&t);      // (only its presence matters to the shape check)
        const long long tiles = (long long)q.B * q.tiles_img;
        if (own) return (q.tiles_img == 1 && conv_small_supported(t, taps, 32)) ? 32 : 0;      // whole groups per 32-channel tile
        if (g_force_bn && conv_small_supported(t, taps, g_force_bn)) return g_force_bn;
        if (taps == 9 && q.TW * q.TH == 128)    // one round of workgroups, or the generic kernel
            return (conv_small_supported(t, 9, 64) && tiles * (q.N / 64) <= 256) ? 64 : 0;
        if (taps == 9) {
            const bool ok64 = conv_small_supported(t, 9, 64), ok32 = conv_small_supported(t, 9, 32);
            if (ok64 && (tiles * (q.N / 64) >= 200 || !ok32)) return 64;
            // (measured, round 3: giving a level that COULD run as multi-tile clusters the cluster's 64 x 64 tile at small batches --
            //  nuScenes 64x2 at 4 images: 66 -> 42 launches per step -- is slower, 84.3 against 86.9 img/s, and neutral for the KITTI
            //  network at 4 / 8 images: a phase costs what a launch costs; the 32-channel tiles' extra workgroups win)
            return ok32 ? 32 : 0;
        }
        const int cand[3] = {32, 64, 128};
        for (int bn : cand)
            if (conv_small_supported(t, taps, bn) && tiles * (q.N / bn) <= 256) return bn;
        return 0;
    }
    // GroupNorm (+ SiLU) folded into the conv's staging: every 1x1, and the 3x3 convs over ONE input tensor (a concatenated
    // input keeps the separate gn_apply launch; rldm_debug_set_flags(131072) keeps it for every 3x3: A/B runs)
    // (round 4) ... and a concatenated 3x3 input on tiles that do not own their image (the 64x4 level's cluster phases, the stand-alone
    // launches of the small-batch configurations): the gn_apply phase / launch it replaces costs 12-23 k cycles, the four-fold repeated
    // arithmetic in the staging ~4.7 k; RLDM_SMALL_CONCAT_GN=0 keeps gn_apply (A/B runs)
    static bool small_concat_gn() { static const bool on = !(getenv("RLDM_SMALL_CONCAT_GN") && atoi(getenv("RLDM_SMALL_CONCAT_GN")) == 0); return on; }
    static bool small_gn_fused(const ConvArgs& a, int taps) {
        if (a.gn == nullptr) return false;
        if (taps == 1) return true;
        if (dbg() & 131072) return false;
        return !a.x1.valid() || (small_concat_gn() && !a.own_image && a.x0.C % 8 == 0 && a.x1.C % 8 == 0);
    }
    bool small_route(const ConvArgs& a, int Cin_t, int R_t, int taps, int Wout, int Hout) const {
        ConvParams q;
        bool epi;
        if (!small_params(a, Cin_t, R_t, taps, Wout, Hout, &q, &epi)) return false;
        return small_bn(q, taps, small_gn_fused(a, taps), a.own_image) != 0;
    }

    int conv_small(const ConvArgs& a, int Cin_t, int R_t, int taps, int Wout, int Hout, Tensor* out) {
        ConvLayer* L = a.layer;
        const int N = L->Cout;
        const bool gn_fused = small_gn_fused(a, taps);         // folded into the conv's staging
        const bool preact = a.gn != nullptr && !gn_fused;      // concatenated 3x3 input: GroupNorm + SiLU in their own launch
        if (a.gn) {
            RLDM_REQUIRE(a.gn->C == Cin_t && Cin_t % a.groups == 0, "conv " + L->name + ": GroupNorm channel mismatch");
            RLDM_REQUIRE(a.x0.P > 0 && (!a.x1.valid() || a.x1.P > 0), "conv " + L->name + ": GroupNorm input without statistics");
        }
        Tensor act;
        if (preact) act = make(a.x0.B, a.x0.W, a.x0.H, Cin_t);
        const Tensor& x0 = preact ? act : a.x0;
        ConvParams p;
        bool epi_res = false;
        RLDM_REQUIRE(small_params(a, Cin_t, R_t, taps, Wout, Hout, &p, &epi_res), "conv " + L->name + ": conv_small route lost");
        p.dbg = dbg();
        p.ts = (getenv("RLDM_TS_ATTN_L") || getenv("RLDM_TS_TRUNK")) ? nullptr : g_ts_buf;      // (the attention timeline owns the buffer then)
        if (getenv("RLDM_TS_ORD")) p.ts = atoi(getenv("RLDM_TS_ORD")) == conv_ord - 1 ? g_ts_buf : nullptr;
        int BN = small_bn(p, taps, gn_fused, a.own_image);
        RLDM_REQUIRE(BN != 0, "conv " + L->name + ": conv_small route lost its instance");
        // (round 5) image-owning tiles of the 32x2 level: 16 channels per workgroup -- the level's persistent launch then runs on 16
        // workgroups per image (all 256 CUs at batch 16) with half the K loop, weight stream and epilogue per workgroup
        if (BN == 32 && a.own_image && own_tile_channels(x0.B, p.TW * p.TH, N) == 16 && !gn_fused && conv_small_supported(p, taps, 16)) BN = 16;
        p.ntile_n = N / BN;

        Tensor y = make(x0.B, Wout, Hout, N);
        if (a.want_stats || a.own_image) add_stats(y, p.tiles_img);
        // normalised copies for the consumers (producer-side GroupNorm): the consumer's input tensor is created by the first
        // producer that writes a part of it and released by the consumer
        std::vector<Tensor> vts;
        if (a.own_image && cur_emit) {
            RLDM_REQUIRE(p.tiles_img == 1 && cur_emit->size() <= 3, "conv " + L->name + ": producer-side GroupNorm lost its tile");
            for (const auto& e : *cur_emit) {
                auto it = view_tensors.find(e.norm);
                if (it == view_tensors.end()) it = view_tensors.emplace(e.norm, make(x0.B, Wout, Hout, e.Ctot)).first;
                vts.push_back(it->second);
            }
        }
        const double fl = 2.0 * (double)x0.B * Wout * Hout * N * ((double)L->Cin * taps + (L->sc_identity ? 0.0 : (double)L->R));
        plan->flops += fl;
        // a phase of the persistent trunk launch (trunk.hip) instead of a launch of its own: the tile owns the image, the input
        // arrives pre-activated (or needs no norm), one of the three instances the trunk kernel carries
        const int cpt_t = Cin_t / 128;
        const int px_t = p.TW * p.TH;
        const int kind_l = (taps == 9 && Cin_t == 256) ? 0 : ((taps == 9 && Cin_t == 512) ? 1 : ((taps == 1 && Cin_t == 256) ? 2 : -1));
        const int trunk_kind = kind_l < 0 ? -1 : kind_l + (BN == 16 ? TK_H16 : (px_t == 32 ? 4 : 0));      // (+4: the 32-pixel instances; +16: 16-channel tiles)
        const int ranks_t = N / BN;
        bool in_trunk = trunk_enabled() && a.own_image && p.tiles_img == 1 && (BN == 32 || BN == 16) && (px_t == 64 || px_t == 32) && !gn_fused &&
                        trunk_kind >= 0 && p.up == 1 && ranks_t >= 2 && ranks_t <= 16 && trunk_grid_fits(ranks_t, x0.B) &&
                        2 * p.TH * (Cin_t / 8) <= 512 && vts.size() <= 2;
        // (a concatenated input is normalised by a gn_apply phase in front of the conv's -- or the conv stays a launch of its own)
        const bool gn_phase_t = in_trunk && preact && Cin_t <= 512 && (Wout * Hout) % ranks_t == 0 && a.x0.C % 8 == 0 &&
                                (!a.x1.valid() || a.x1.C % 8 == 0) && !(dbg() & (1 << 27));
        in_trunk = in_trunk && (!preact || gn_phase_t);
        // ... or a phase of a MULTI-TILE cluster (the 64x4 level at batch <= 16): conv_small's default 64-pixel x 64-channel tiles, the
        // consumer-side GroupNorm fold stays inside the phase
        // (3x3 over 128 channels -- the all-taps-ring instance, 211 registers on its own -- does not fit beside the phase loop's state)
        const int kind_c = (taps == 9 && Cin_t == 128 && !getenv("RLDM_NO_CL128")) ? TK_CL_3x3_128 : (taps == 9 && Cin_t == 256) ? TK_CL_3x3_256 :
                           (taps == 9 && Cin_t == 384) ? TK_CL_3x3_384 : (taps == 9 && Cin_t == 512) ? TK_CL_3x3_512 :
                           (taps == 1 && Cin_t == 256) ? TK_CL_1x1_256 : -1;
        const int ranks_c = cluster_ranks(x0.B, N, Wout * Hout);
        const bool in_cluster = !in_trunk && cluster_enabled() && !(dbg() & 512) && !a.own_image && kind_c >= 0 && BN == 64 && px_t == 64 &&
                                (p.up == 1 || p.up == 2) && vts.empty() && ranks_c == (N / 64) * p.tiles_img &&
                                p.Win * p.up == Wout && p.Hin * p.up == Hout;
        // GroupNorm + SiLU once, ahead of the conv: every channel tile of the conv would otherwise redo it (4-8x at these levels) --
        // as a launch (norm.hip) or, in front of a multi-tile cluster phase, as a phase of the same persistent launch
        if (preact) {
            const bool gn_phase = gn_phase_t || (in_cluster && Cin_t <= 512 && (Wout * Hout) % ranks_c == 0 && a.x0.C % 8 == 0 &&
                                                 (!a.x1.valid() || a.x1.C % 8 == 0) && !(dbg() & (1 << 27)));
            if (gn_phase_t) trunk_begin(x0.B, ranks_t);
            else if (gn_phase) trunk_begin(x0.B, ranks_c, N / 64, 2);
            else note_launch();
            if (!dry) {
                GnApplyParams g;
                memset(&g, 0, sizeof(g));
                g.x0 = tptr(a.x0); g.x1 = tptr(a.x1);
                g.C0 = a.x0.C; g.C1 = a.x1.valid() ? a.x1.C : 0;
                g.st0 = sptr(a.x0); g.st1 = sptr(a.x1);
                g.P0 = a.x0.P; g.P1 = a.x1.valid() ? a.x1.P : 0;
                g.B = a.x0.B; g.npix = a.x0.W * a.x0.H;
                g.groups = a.groups;
                g.gamma = a.gn->gamma.as<float>(); g.beta = a.gn->beta.as<float>();
                g.eps = a.eps; g.silu = a.silu;
                g.y = tptr(act);
                const double by = (double)g.B * g.npix * Cin_t * 4.0;
                if (gn_phase) {
                    TrunkPhase ph;
                    memset(&ph, 0, sizeof(ph));
                    auto put64 = [&](int at, const void* ptr) {
                        const unsigned long long u = (unsigned long long)(uintptr_t)ptr;
                        ph.w[at] = (unsigned)u;
                        ph.w[at + 1] = (unsigned)(u >> 32);
                    };
                    put64(TW_X0, g.x0); put64(TW_R0, g.x1); put64(TW_ST0, g.st0); put64(TW_RES, g.st1);
                    put64(TW_GAMMA, g.gamma); put64(TW_BETA, g.beta); put64(TW_Y, g.y);
                    ph.w[TW_R0C] = g.C0; ph.w[TW_R1C] = g.C1; ph.w[TW_P0] = g.P0; ph.w[TW_TILES_H] = g.P1; ph.w[TW_WIN] = g.npix;
                    ph.w[TW_GROUPS] = g.groups; ph.w[TW_SILU] = g.silu;
                    const float inv_n = (float)(1.0 / ((double)g.npix * (Cin_t / g.groups)));
                    memcpy(&ph.w[TW_INVN], &inv_n, 4); memcpy(&ph.w[TW_EPS], &g.eps, 4);
                    ph.w[TW_KIND] = TK_GN_APPLY; ph.w[TW_TEMBOFF] = (unsigned)-1;
                    pend.phases.push_back(ph);
                    pend.lds = std::max(pend.lds, (size_t)(2 * 512 * 8 + 2 * 512 * 4));
                    pend.bytes += by;
                    pend.standalone.push_back({[g](hipStream_t s) { return launch_gn_apply(g, s); }, "gn_apply_kernel", 0.0, by});
                } else {
                    plan->ops.push_back({[g](hipStream_t s) { return launch_gn_apply(g, s); }, "gn_apply_kernel", 0.0, by});
                }
            }
        }
        if (in_trunk) trunk_begin(x0.B, ranks_t);
        else if (in_cluster) trunk_begin(x0.B, ranks_c, N / 64, 2);
        else note_launch();
        in_trunk = in_trunk || in_cluster;
        (void)cpt_t;
        if (!dry) {
            ConvLayer::Packed* pk = nullptr;
            if (BN == 16 ? L->get_fragpacked16(Cin_t, epi_res, &pk) : L->get_fragpacked(Cin_t, conv_small_kgroups(BN), epi_res, &pk)) return 1;
            p.x0 = tptr(x0);
            p.r0 = epi_res ? nullptr : tptr(a.r0);
            p.r1 = epi_res ? nullptr : tptr(a.r1);
            p.res = epi_res ? tptr(a.r0) : nullptr;
            p.wpk = pk->w.as<bf16_t>();
            p.bias = pk->bias.as<float>();
            if (gn_fused) {
                p.st0 = sptr(a.x0);
                p.P0 = a.x0.P;
                p.gn_gamma = a.gn->gamma.as<float>();
                p.gn_beta = a.gn->beta.as<float>();
                if (p.C1 != 0) {                // concatenated input: the second tensor and its statistics
                    p.x1 = tptr(a.x1);
                    p.st1 = sptr(a.x1);
                    p.P1 = a.x1.P;
                }
            }
            p.y = tptr(y);
            p.y_ld = N;
            p.y_stats = y.P ? ptr<float2>(y.st_off) : nullptr;
            p.temb_ld = temb_ld;
            for (size_t v = 0; v < vts.size(); ++v) {
                const auto& e = (*cur_emit)[v];
                NormView& nv = p.nv[v];
                nv.y = tptr(vts[v]) + e.ch_off;
                nv.gamma = e.norm->gamma.as<float>() + e.ch_off;
                nv.beta = e.norm->beta.as<float>() + e.ch_off;
                nv.ld = e.Ctot;
                const int cpg = e.Ctot / e.groups;
                nv.cpg_shift = 0;
                while ((1 << nv.cpg_shift) < cpg) ++nv.cpg_shift;
                nv.inv_n = (float)(1.0 / ((double)Wout * Hout * cpg));
                nv.eps = e.eps;
                nv.silu = e.silu;
            }
            p.nviews = (int)vts.size();
            Plan* pl = plan;
            const int temb_off = a.temb_off;
            const double by = (double)x0.B * x0.W * x0.H * Cin_t * 2.0 + (double)N * (L->Cin * taps + L->R) * 2.0 +
                              (double)x0.B * Wout * Hout * N * 2.0 * (1.0 + (double)vts.size()) + (double)x0.B * Wout * Hout * R_t * 2.0;
            const std::string kname = "conv_small_kernel<" + std::to_string(p.TW * p.TH) + "," + std::to_string(BN) + ",taps" + std::to_string(taps) + ">";
            if (in_trunk) {
                TrunkPhase ph;
                memset(&ph, 0, sizeof(ph));
                auto put64 = [&](int at, const void* ptr) {
                    const unsigned long long u = (unsigned long long)(uintptr_t)ptr;
                    ph.w[at] = (unsigned)u;
                    ph.w[at + 1] = (unsigned)(u >> 32);
                };
                auto putf = [&](int at, float f) { memcpy(&ph.w[at], &f, 4); };
                put64(TW_X0, p.x0); put64(TW_R0, p.r0); put64(TW_R1, p.r1); put64(TW_WPK, p.wpk); put64(TW_BIAS, p.bias);
                put64(TW_Y, p.y); put64(TW_YSTATS, p.y_stats); put64(TW_RES, p.res);
                ph.w[TW_R0C] = p.R0; ph.w[TW_R1C] = p.R1; ph.w[TW_WIN] = p.Win; ph.w[TW_HIN] = p.Hin; ph.w[TW_WOUT] = p.Wout;
                ph.w[TW_HOUT] = p.Hout; ph.w[TW_TW] = p.TW; ph.w[TW_TH] = p.TH; ph.w[TW_COLB] = p.colb; ph.w[TW_THSHIFT] = p.th_shift;
                ph.w[TW_N] = p.N; ph.w[TW_YLD] = p.y_ld; ph.w[TW_NVIEWS] = p.nviews;
                if (in_cluster) {
                    const int KG = 4, cpt = Cin_t / (16 * KG);
                    const int TPG = taps == 1 ? 1 : (cpt <= 2 ? 9 : (cpt <= 4 ? 3 : 1));      // conv_small_body.h, MI == 2
                    ph.w[TW_KIND] = kind_c;
                    ph.w[TW_G] = std::min(TPG * cpt, RLDM_TRUNK_PREFETCH);
                    ph.w[TW_NMINE] = taps * cpt + ((p.R0 + p.R1) / 16) / KG;
                    ph.w[TW_WBYTES] = (unsigned)((N / 32) * KG * ph.w[TW_NMINE]) * 1024u;
                    RLDM_REQUIRE(p.nviews == 0, "conv " + L->name + ": a multi-tile cluster phase writes no views");
                    put64(TW_ST0, p.st0); put64(TW_GAMMA, p.gn_gamma); put64(TW_BETA, p.gn_beta);
                    ph.w[TW_P0] = p.P0; ph.w[TW_GROUPS] = p.gn_groups; ph.w[TW_MAGIC_CPG] = p.magic_cpg;
                    putf(TW_INVN, p.gn_inv_n); putf(TW_EPS, p.gn_eps);
                    ph.w[TW_SILU] = p.silu; ph.w[TW_TILES_H] = p.tiles_h; ph.w[TW_TILES_IMG] = p.tiles_img;
                    ph.w[TW_UP] = p.up;
                    put64(TW_X1, p.x1); put64(TW_ST1, p.st1);
                    ph.w[TW_C0] = p.C0; ph.w[TW_C1] = p.C1; ph.w[TW_P1] = p.P1;
                } else {
                const int KG = 8, ksc = BN == 16 ? 32 : 16, cpt = Cin_t / (ksc * KG);
                const int G = BN == 16 ? (taps == 9 ? 9 : 1) * cpt : ((trunk_kind & 3) == 0 ? 18 : ((trunk_kind & 3) == 1 ? 12 : cpt));
                ph.w[TW_KIND] = trunk_kind;
                ph.w[TW_G] = std::min(G, RLDM_TRUNK_PREFETCH);       // == kTrunkPrefetch (conv_small_body.h)
                ph.w[TW_NMINE] = taps * cpt + ((p.R0 + p.R1) / ksc) / KG;
                ph.w[TW_WBYTES] = (unsigned)((N / BN) * KG * ph.w[TW_NMINE]) * 1024u;
                }
                ph.w[TW_TEMBOFF] = (unsigned)temb_off;
                for (int v = 0; v < p.nviews; ++v) {
                    const int at = TW_NV0 + v * TW_NVSTRIDE;
                    put64(at, p.nv[v].y); put64(at + 2, p.nv[v].gamma); put64(at + 4, p.nv[v].beta);
                    ph.w[at + 6] = p.nv[v].ld; ph.w[at + 7] = p.nv[v].cpg_shift;
                    putf(at + 8, p.nv[v].inv_n); putf(at + 9, p.nv[v].eps);
                    ph.w[at + 10] = p.nv[v].silu;
                }
                pend.phases.push_back(ph);
                pend.lds = std::max(pend.lds, conv_small_lds_bytes(p, taps, BN));
                pend.flops += fl;
                pend.bytes += by;
            }
            Op standalone{[p, BN, taps, pl, temb_off](hipStream_t s) mutable {
                if (temb_off >= 0) {
                    p.temb = pl->io.temb + temb_off;
                    p.step_ptr = pl->io.step_ptr;
                    p.temb_rows_per_step = pl->io.temb_rows_per_step;
                    p.temb_per_sample = pl->io.temb_per_sample;
                }
                return launch_conv_small(p, taps, BN, s);
            }, kname, fl, by};
            if (in_trunk) pend.standalone.push_back(standalone);
            else plan->ops.push_back(standalone);
        }
        if (preact) release(act);
        *out = y;
        return 0;
    }

    // conv_stream.hip route: 3x3 / stride 1 convs whose output has at least 128 tiles of 32 x 8 pixels x 128 channels, or
    // (the 128x8 level) of 16 x 8 pixels x 64 channels
    static inline const int kSubMinBlocks = getenv("RLDM_SUB_MIN") ? atoi(getenv("RLDM_SUB_MIN")) : 96;     // (env: tuning runs)
    static bool stream_params_tw(const ConvArgs& a, int Cin_t, int R_t, int taps, int Wout, int Hout, int TW, long long min_blocks,
                                 long long max_blocks, ConvParams* q, int TH = 8, int inst = 0) {
        if (dbg() & 2048) return false;
        if (taps != 9 || (a.stride != 1 && !(a.stride == 2 && inst == 5)) || a.pad_mode != 0 || a.out_f32_nchw || g_force_bm) return false;
        // (inst 6, round 4: nearest x2 + 3x3 in its sub-pixel form -- the tiles are INPUT tiles, four parity workgroups each)
        const bool sub = inst == 6;
        if (sub) {
            if (a.up != 2 || R_t != 0 || Wout != 2 * a.x0.W || Hout != 2 * a.x0.H) return false;
            Wout = a.x0.W; Hout = a.x0.H;
        }
        if (Wout % TW != 0 || Hout % TH != 0) return false;
        memset(q, 0, sizeof(*q));
        q->C0 = a.x0.C;
        q->C1 = Cin_t - a.x0.C;
        q->R0 = a.r0.valid() ? a.r0.C : 0;
        q->R1 = R_t - q->R0;
        q->B = a.x0.B; q->Win = a.x0.W; q->Hin = a.x0.H;
        q->up = sub ? 1 : a.up; q->stride = a.stride; q->pad_lo = 1;
        q->Wout = sub ? 2 * Wout : Wout; q->Hout = sub ? 2 * Hout : Hout;
        q->TW = TW; q->TH = TH; q->th_shift = TH == 16 ? 4 : (TH == 8 ? 3 : 2);
        q->st_inst = inst;
        ConvTile t;
        t.BM = 256; t.BN = 128; t.CK = 64; t.taps = 9;
        q->colb = conv_halo_col_bytes(t, TH, a.stride);
        q->tiles_h = Hout / TH;
        q->tiles_img = (Wout / TW) * q->tiles_h;
        {
            const int thv = inst == 7 ? TH : (TH - 1) * a.stride + 3;       // (7: only the tile's own rows are staged)
            q->magic_thv = ((1 << 20) + thv - 1) / thv;
        }
        const int cpg = std::max(1, Cin_t / a.groups);
        q->magic_cpg = ((1 << 20) + cpg - 1) / cpg;
        q->gn_inv_n = (float)(1.0 / ((double)a.x0.W * a.x0.H * cpg));
        q->N = a.layer->Cout;
        q->silu = a.silu;
        q->gn_eps = a.eps;
        q->gn_groups = a.groups;
        q->ksplit = 1;
        if (a.gn) q->st0 = reinterpret_cast<const float2*>(q);      // (only its presence matters to the shape check)
        const long long blocks = (long long)q->B * q->tiles_img * (q->N / conv_stream_bn(*q)) * (sub ? 4 : 1);
        const bool ok = conv_stream_supported(*q, 9) && ((dbg() & 4096) || (blocks >= min_blocks && blocks <= max_blocks));
        q->st0 = nullptr;
        return ok;
    }
    static bool stream_params(const ConvArgs& a, int Cin_t, int R_t, int taps, int Wout, int Hout, ConvParams* q) {
        // 256-pixel tiles when they fill the chip; else the 4-k-group instance (it re-streams the weights per 128 pixels:
        // only where its grid is about one or two rounds); else 256-pixel tiles on at least half the chip
        // (round 4) half-size workgroups, two per CU: 16 x 8 tiles x 128 channels on 4 waves where that grid is at least ~1.5 per CU
        // (UNet 256x16 level at batch >= 12, the VAE decoder's 128 / 256-channel levels), x 64 channels x 2 k-groups for the 128x8
        // level and the VAE's 64-channel level
        const int N_ = a.layer->Cout;
        // (round 4) stride 2 (Downsample2D, pad 1) on the 64-pixel x 128-channel tile with a 17 x 17 halo: the 256x16 -> 128x8 down-sampler ran on
        // the generic kernel's half-empty 256-pixel tile; rldm_debug_set_flags2(128) keeps it there
        if (a.stride == 2) {
            if ((dbg2() & 128) || N_ % 128 != 0 || R_t != 0 || a.up != 1) return false;
            if (stream_params_tw(a, Cin_t, R_t, taps, Wout, Hout, 8, kInst4MinBlocks, 512, q, 8, 5)) return true;
            // outputs of 4 beams (the 128x8 -> 64x4 down-sampler): 16 x 4 tiles, from 48 workgroups on
            return Hout == 4 && !getenv("RLDM_NO_S2_H4") && stream_params_tw(a, Cin_t, R_t, taps, Wout, Hout, 16, 48, 512, q, 4, 5);
        }
        // (experiment, rldm_debug_set_flags2(32)) the 256 x 128 tile with specialised waves wherever the 8-wave 256 x 128 instance would run
        if ((dbg2() & 32) && N_ % 128 == 0 && stream_params_tw(a, Cin_t, R_t, taps, Wout, Hout, 32, 200, 1ll << 40, q, 8, 3)) return true;
        // (tests, rldm_debug_set_flags2(64)) the 64-pixel x 128-channel tile first, at any level it fits
        if ((dbg2() & 64) && N_ % 128 == 0 && stream_params_tw(a, Cin_t, R_t, taps, Wout, Hout, 8, 192, 512, q, 8, 4)) return true;
        // (round 4) nearest x2 + 3x3 as four 2x2 convs over the input (sub-pixel form: 4 taps instead of 9 per output pixel) on the 4-wave
        // 128 x 128 tile; rldm_debug_set_flags2(1 << 27) keeps the 3x3 over the up-sampled halo
        if (!(dbg2() & (1 << 27)) && !(dbg2() & 1) && a.up == 2 && N_ % 128 == 0 &&
            (stream_params_tw(a, Cin_t, R_t, taps, Wout, Hout, 16, kSubMinBlocks, 1ll << 40, q, 8, 6) ||
             (a.x0.H % 8 != 0 && stream_params_tw(a, Cin_t, R_t, taps, Wout, Hout, 32, kSubMinBlocks, 1ll << 40, q, 4, 6)))) return true;   // (inputs of 4 beams: 32 x 4 tiles)
        // (round 4) images of 16 beams: 8 x 16 tiles as tall as the image -- five staged pieces per thread instead of six; rldm_debug_set_flags2(1 << 29):
        // the 16 x 8 tiles
        if (!(dbg2() & 1) && !(dbg2() & (1 << 29)) && N_ % 128 == 0 && (Hout == 16 || Hout == 8) && a.up == 1 &&
            stream_params_tw(a, Cin_t, R_t, taps, Wout, Hout, Hout == 16 ? 8 : 16, 384, 1ll << 40, q, Hout, 7)) return true;       // (8 beams: 16 x 8)
        if (!(dbg2() & 1) && N_ % 128 == 0 && stream_params_tw(a, Cin_t, R_t, taps, Wout, Hout, 16, 384, 1ll << 40, q, 8, 1)) return true;
        if (!(dbg2() & 4) && N_ % 128 != 0 && stream_params_tw(a, Cin_t, R_t, taps, Wout, Hout, 16, 384, 1ll << 40, q, 8, 2)) return true;
        if (stream_params_tw(a, Cin_t, R_t, taps, Wout, Hout, 32, 200, 1ll << 40, q)) return true;
        // (round 4) the 128x8 level: 64-pixel x 128-channel x 2-k-group tiles (8 x 8: a smaller halo, normalised once for all 128 channels,
        // half the partial sums to exchange); rldm_debug_set_flags2(16) keeps the 128 x 64 x 4-k-group tiles
        if (!(dbg2() & 16) && !(dbg() & 16384) && N_ % 128 == 0 && Hout == 8 &&       // (at the 256x16 level of small batches it breaks the clusters: -4 %; RangeDM at batch 1: +1 %, not worth a rule)
            stream_params_tw(a, Cin_t, R_t, taps, Wout, Hout, 8, kInst4MinBlocks, 320, q, 8, 4)) return true;      // (one round of 8-wave workgroups: at 512
        // blocks -- the 256-channel up-sampler conv of the level -- two co-resident 4-wave workgroups per CU win, 23.6 against 27.1 us)
        if (!(dbg2() & 2) && !(dbg() & 16384) && stream_params_tw(a, Cin_t, R_t, taps, Wout, Hout, 16, 257, 512, q, 8, 2)) return true;
        if (!(dbg() & 16384) && stream_params_tw(a, Cin_t, R_t, taps, Wout, Hout, 16, 128, 512, q)) return true;
        // images of 4 beams (nuScenes' 128 x 4 level at batch 32): the same 128-pixel instance on 32 x 4 tiles (round 3; it ran on the
        // generic kernel at 27.8 us / 257 TFLOP/s per conv: 16 % of that configuration's step)
        if (Hout % 8 != 0 && !(dbg() & 16384) && !(dbg() & 16) && stream_params_tw(a, Cin_t, R_t, taps, Wout, Hout, 32, 128, 512, q, 4)) return true;
        return stream_params_tw(a, Cin_t, R_t, taps, Wout, Hout, 32, 128, 1ll << 40, q);
    }

    // (round 5) the conv pairs of the 128x8 level -- 64-pixel x 128-channel tiles, 16 workgroups per image -- as 2-phase persistent launches
    // (trunk variant 5); RLDM_STREAM_PAIRS64=0 / 1 overrides the default
    static bool stream_pairs64() { static const bool on = getenv("RLDM_STREAM_PAIRS64") ? atoi(getenv("RLDM_STREAM_PAIRS64")) != 0 : kStreamPairs64Default; return on; }
    int conv_stream(const ConvArgs& a, int Cin_t, int R_t, int Wout, int Hout, Tensor* out) {
        ConvLayer* L = a.layer;
        const int N = L->Cout;
        const Tensor& x0 = a.x0;
        ConvParams p;
        RLDM_REQUIRE(stream_params(a, Cin_t, R_t, 9, Wout, Hout, &p), "conv " + L->name + ": conv_stream route lost");
        if (a.gn) {
            RLDM_REQUIRE(a.gn->C == Cin_t && Cin_t % a.groups == 0, "conv " + L->name + ": GroupNorm channel mismatch");
            RLDM_REQUIRE(x0.P > 0 && (!a.x1.valid() || a.x1.P > 0), "conv " + L->name + ": GroupNorm input without statistics");
        }
        p.dbg = dbg();
        p.ts = (getenv("RLDM_TS_ATTN_L") || getenv("RLDM_TS_TRUNK")) ? nullptr : g_ts_buf;      // (the attention timeline owns the buffer then)
        if (getenv("RLDM_TS_ORD")) p.ts = atoi(getenv("RLDM_TS_ORD")) == conv_ord - 1 ? g_ts_buf : nullptr;   // ONE conv of a network
        const bool sub = p.st_inst == 6;
        p.ntile_n = N / conv_stream_bn(p) * (sub ? 4 : 1);      // (grid x: channel tiles x parities)
        p.exp = (dbg2() >> 8) & 255;            // (bits 8..15 only: 16..23 are trunk variant 4's start offset)
        if (p.exp & 2) {
            static int* locks = nullptr;            // (experiment: never freed)
            if (!locks) {
                RLDM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&locks), 4096 * sizeof(int)));
                RLDM_HIP_CHECK(hipMemset(locks, 0, 4096 * sizeof(int)));
            }
            p.cu_lock = locks;
        }
        Tensor y = make(x0.B, Wout, Hout, N);
        if (a.want_stats) add_stats(y, p.tiles_img * (sub ? 4 : 1));
        const double fl_ref = 2.0 * (double)x0.B * Wout * Hout * N * ((double)L->Cin * 9 + (L->sc_identity ? 0.0 : (double)L->R));
        plan->flops += fl_ref;                  // (the network's nominal count: 9 taps)
        // what THIS launch / phase executes (the bench line's roofline is priced with it): the sub-pixel form multiplies 4 taps per output pixel
        const double fl = sub ? fl_ref * 4.0 / 9.0 : fl_ref;
        // a phase of the persistent launch (trunk.hip, variants 2 / 3): the image's tiles_img x ntile_n workgroups (16 at both
        // full-resolution levels) form a cluster on one XCD; consecutive convs of a level hand over through its L2 -- no end-of-kernel
        // write-back of the 16.8 MB outputs, no argument fetch / cold start per layer.  rldm_debug_set_flags(1 << 28): separate launches
        const int ranks_s = p.tiles_img * p.ntile_n;            // (sub-pixel form: input tiles x parities, one 128-channel tile)
        // (round 4) the 4-wave 128 x 128 instance: 32 workgroups per image, two per CU -- trunk variant 4; rldm_debug_set_flags2(8): launches
        const bool inst1 = p.st_inst == 1 || p.st_inst == 7;      // (the 4-wave 128 x 128 tile: 16 x 8, or 8 x 16 as tall as the image)
        const int per_cu = inst1 || sub ? 2 : 1;
        const bool in_stream_cluster = cluster_enabled() && !(dbg() & (1 << 28)) && ranks_s >= 2 && ranks_s <= 16 * per_cu &&
                                       trunk_grid_fits(ranks_s, x0.B, per_cu) && y.P <= kFoldAboveP && N % 128 == 0 &&
                                       ((p.st_inst == 4 && stream_pairs64()) ||       // (round 5: the 128x8 level's 64-pixel x 128-channel tile as phases)
                                        (p.st_inst == 0 && (p.TW * p.TH == 256 || (dbg() & (1 << 30)))) ||     // (the 128x8 level's conv
                                        // PAIRS measured slower as 2-phase launches than as two launches, 216.9 against 220.0 img/s: off unless 1 << 30)
                                        ((inst1 || (sub && N == 128 && p.TH == 8 && !(dbg2() & (1 << 28)))) && !(dbg2() & 8) && conv_stream_lds_bytes(p) <= 80 * 1024));
        if (in_stream_cluster) trunk_begin(x0.B, ranks_s, sub ? 1 : p.ntile_n, inst1 || sub || p.TW * p.TH == 256 || p.st_inst == 4 ? 4 : 2,
                                           inst1 || sub ? 4 : (p.st_inst == 4 ? 5 : (p.TW * p.TH == 256 ? 2 : 3)));
        else note_launch();
        if (!dry) {
            ConvLayer::Packed* pk = nullptr;
            if (sub ? L->get_subpixpacked(Cin_t, &pk) : L->get_streampacked(Cin_t, conv_stream_kgroups(p), &pk)) return 1;
            p.x0 = tptr(x0);
            p.x1 = tptr(a.x1);
            p.r0 = tptr(a.r0);
            p.r1 = tptr(a.r1);
            p.wpk = pk->w.as<bf16_t>();
            p.bias = pk->bias.as<float>();
            if (a.gn) {
                p.st0 = sptr(x0);
                p.st1 = sptr(a.x1);
                p.P0 = x0.P;
                p.P1 = a.x1.valid() ? a.x1.P : 0;
                p.gn_gamma = a.gn->gamma.as<float>();
                p.gn_beta = a.gn->beta.as<float>();
            }
            p.y = tptr(y);
            p.y_ld = N;
            p.y_stats = y.P ? ptr<float2>(y.st_off) : nullptr;
            p.temb_ld = temb_ld;
            Plan* pl = plan;
            const int temb_off = a.temb_off;
            const double by = (double)x0.B * x0.W * x0.H * Cin_t * 2.0 + (double)N * (L->Cin * 9 + L->R) * 2.0 +
                              (double)x0.B * Wout * Hout * N * 2.0 + (double)x0.B * Wout * Hout * R_t * 2.0;
            if (in_stream_cluster) {
                TrunkPhase ph;
                memset(&ph, 0, sizeof(ph));
                auto put64 = [&](int at, const void* ptr) {
                    const unsigned long long u = (unsigned long long)(uintptr_t)ptr;
                    ph.w[at] = (unsigned)u;
                    ph.w[at + 1] = (unsigned)(u >> 32);
                };
                auto putf = [&](int at, float f) { memcpy(&ph.w[at], &f, 4); };
                put64(TW_X0, p.x0); put64(TW_X1, p.x1); put64(TW_R0, p.r0); put64(TW_R1, p.r1); put64(TW_WPK, p.wpk); put64(TW_BIAS, p.bias);
                put64(TW_Y, p.y); put64(TW_YSTATS, p.y_stats); put64(TW_ST0, p.st0); put64(TW_ST1, p.st1);
                put64(TW_GAMMA, p.gn_gamma); put64(TW_BETA, p.gn_beta);
                ph.w[TW_C0] = p.C0; ph.w[TW_C1] = p.C1; ph.w[TW_R0C] = p.R0; ph.w[TW_R1C] = p.R1; ph.w[TW_P0] = p.P0; ph.w[TW_P1] = p.P1;
                ph.w[TW_WIN] = p.Win; ph.w[TW_HIN] = p.Hin; ph.w[TW_WOUT] = p.Wout; ph.w[TW_HOUT] = p.Hout; ph.w[TW_UP] = p.up;
                ph.w[TW_TW] = p.TW; ph.w[TW_TH] = p.TH; ph.w[TW_COLB] = p.colb; ph.w[TW_THSHIFT] = p.th_shift;
                ph.w[TW_TILES_H] = p.tiles_h; ph.w[TW_TILES_IMG] = p.tiles_img; ph.w[TW_MAGIC_THV] = p.magic_thv;
                ph.w[TW_MAGIC_CPG] = p.magic_cpg; ph.w[TW_GROUPS] = p.gn_groups; ph.w[TW_SILU] = p.silu;
                putf(TW_INVN, p.gn_inv_n); putf(TW_EPS, p.gn_eps);
                ph.w[TW_N] = p.N; ph.w[TW_YLD] = p.y_ld;
                ph.w[TW_KIND] = TK_STREAM; ph.w[TW_TEMBOFF] = (unsigned)temb_off; ph.w[TW_SUB] = sub ? 1 : (p.st_inst == 7 ? 2 : 0);
                pend.phases.push_back(ph);
                pend.lds = std::max(pend.lds, conv_stream_lds_bytes(p));
                pend.flops += fl;
                pend.bytes += by;
            }
            Op standalone{[p, pl, temb_off](hipStream_t s) mutable {
                if (temb_off >= 0) {
                    p.temb = pl->io.temb + temb_off;
                    p.step_ptr = pl->io.step_ptr;
                    p.temb_rows_per_step = pl->io.temb_rows_per_step;
                    p.temb_per_sample = pl->io.temb_per_sample;
                }
                return launch_conv_stream(p, s);
            }, "conv_stream_kernel<" + std::to_string(p.TW * p.TH) + "," + std::to_string(conv_stream_bn(p)) + ",CK64,taps9" + (p.stride == 2 ? ",s2" : "") + (p.st_inst == 6 ? ",sub" : "") + ">", fl, by};
            if (in_stream_cluster) pend.standalone.push_back(standalone);
            else plan->ops.push_back(standalone);
        }
        *out = y;
        return 0;
    }

    // conv_regw.hip, conv_c16_kernel (round 4): the network's input layer (16 padded input channels, 128 | N, no norm / residual / time
    // embedding): every weight in one wave's registers; RLDM_NO_C16=1 keeps the generic kernel
    static bool c16_params(const ConvArgs& a, int Cin_t, int R_t, int taps, int Wout, int Hout, ConvParams* q) {
        static const bool off = getenv("RLDM_NO_C16") != nullptr;
        if (off || (dbg() & 2048) || g_force_bm || taps != 9 || a.stride != 1 || a.pad_mode != 0 || a.up != 1 || a.out_f32_nchw) return false;
        if (a.x1.valid() || a.gn || a.temb_off >= 0 || R_t != 0 || Cin_t != 16 || a.layer->Cout % 128 != 0 || Wout % 16 != 0 || Hout % 8 != 0) return false;
        memset(q, 0, sizeof(*q));
        q->C0 = 16;
        q->B = a.x0.B; q->Win = a.x0.W; q->Hin = a.x0.H;
        q->up = 1; q->stride = 1; q->pad_lo = 1;
        q->Wout = Wout; q->Hout = Hout;
        q->TW = 16; q->TH = 8; q->th_shift = 3;
        q->tiles_h = Hout / 8;
        q->tiles_img = (Wout / 16) * q->tiles_h;
        q->N = a.layer->Cout;
        q->ntile_n = q->N / 128;
        q->ksplit = 1;
        return conv_c16_supported(*q) && (long long)q->B * q->tiles_img * q->ntile_n >= 64;
    }
    int conv_c16(const ConvArgs& a, int Wout, int Hout, Tensor* out) {
        ConvLayer* L = a.layer;
        const int N = L->Cout;
        const Tensor& x0 = a.x0;
        ConvParams p;
        RLDM_REQUIRE(c16_params(a, x0.C, 0, 9, Wout, Hout, &p), "conv " + L->name + ": conv_c16 route lost");
        Tensor y = make(x0.B, Wout, Hout, N);
        if (a.want_stats) add_stats(y, p.tiles_img);
        const double fl = 2.0 * (double)x0.B * Wout * Hout * N * (double)L->Cin * 9;
        plan->flops += fl;
        note_launch();
        if (!dry) {
            ConvLayer::Packed* pk = nullptr;
            if (L->get_c16packed(&pk)) return 1;
            p.x0 = tptr(x0);
            p.wpk = pk->w.as<bf16_t>();
            p.bias = pk->bias.as<float>();
            p.y = tptr(y);
            p.y_ld = N;
            p.y_stats = y.P ? ptr<float2>(y.st_off) : nullptr;
            const double by = (double)x0.B * x0.W * x0.H * 16 * 2.0 + (double)N * L->Cin * 9 * 2.0 + (double)x0.B * Wout * Hout * N * 2.0;
            Plan* pl = plan;
            const bool first_of_step = a.first_of_step;
            plan->ops.push_back({[p, pl, first_of_step](hipStream_t s) mutable {
                p.step_inc = (first_of_step && pl->io.pack_fused) ? pl->io.step_inc : nullptr;
                return launch_conv_c16(p, s);
            }, "conv_c16_kernel<128,128,taps9>", fl, by});
        }
        *out = y;
        return 0;
    }

    // conv_regw.hip, conv_o4_kernel (round 4): the UNet's output layer (GroupNorm + SiLU -> 3x3 over 128 channels -> <= 4 channels, fp32 NCHW + the
    // scheduler step); RLDM_NO_O4=1 keeps the generic kernel
    static bool o4_params(const ConvArgs& a, int Cin_t, int R_t, int taps, int Wout, int Hout, ConvParams* q) {
        static const bool off = getenv("RLDM_NO_O4") != nullptr;
        if (off || (dbg() & 2048) || g_force_bm || taps != 9 || a.stride != 1 || a.pad_mode != 0 || a.up != 1 || !a.out_f32_nchw) return false;
        if (a.x1.valid() || a.temb_off >= 0 || R_t != 0 || Cin_t != 128 || a.layer->Cin != 128 || a.layer->Cout > 4 || Wout % 16 != 0 || Hout % 8 != 0) return false;
        memset(q, 0, sizeof(*q));
        q->C0 = 128;
        q->B = a.x0.B; q->Win = a.x0.W; q->Hin = a.x0.H;
        q->up = 1; q->stride = 1; q->pad_lo = 1;
        q->Wout = Wout; q->Hout = Hout;
        q->TW = 16; q->TH = 8; q->th_shift = 3;
        q->tiles_h = Hout / 8;
        q->tiles_img = (Wout / 16) * q->tiles_h;
        const int cpg = std::max(1, Cin_t / a.groups);
        q->gn_inv_n = (float)(1.0 / ((double)a.x0.W * a.x0.H * cpg));
        q->N = a.layer->Cout;
        q->ntile_n = 1;
        q->silu = a.silu;
        q->gn_eps = a.eps;
        q->gn_groups = a.groups;
        q->ksplit = 1;
        if (a.gn) q->st0 = reinterpret_cast<const float2*>(q);      // (only their presence matters to the shape check)
        q->y_nchw = reinterpret_cast<float*>(q);
        const bool ok = conv_o4_supported(*q) && (long long)q->B * q->tiles_img >= 64;
        q->st0 = nullptr;
        q->y_nchw = nullptr;
        return ok;
    }
    int conv_o4(const ConvArgs& a, int Wout, int Hout, Tensor* out) {
        ConvLayer* L = a.layer;
        const int N = L->Cout;
        const Tensor& x0 = a.x0;
        ConvParams p;
        RLDM_REQUIRE(o4_params(a, x0.C, 0, 9, Wout, Hout, &p), "conv " + L->name + ": conv_o4 route lost");
        if (a.gn) {
            RLDM_REQUIRE(a.gn->C == 128 && 128 % a.groups == 0, "conv " + L->name + ": GroupNorm channel mismatch");
            RLDM_REQUIRE(x0.P > 0, "conv " + L->name + ": GroupNorm input without statistics");
        }
        const double fl = 2.0 * (double)x0.B * Wout * Hout * N * (double)L->Cin * 9;
        plan->flops += fl;
        note_launch();
        if (!dry) {
            ConvLayer::Packed* pk = nullptr;
            if (L->get_streampacked(128, 2, &pk)) return 1;
            p.x0 = tptr(x0);
            p.wpk = pk->w.as<bf16_t>();
            p.bias = pk->bias.as<float>();
            if (a.gn) {
                p.st0 = sptr(x0);
                p.P0 = x0.P;
                p.gn_gamma = a.gn->gamma.as<float>();
                p.gn_beta = a.gn->beta.as<float>();
            }
            const double by = (double)x0.B * x0.W * x0.H * 128 * 2.0 + (double)N * L->Cin * 9 * 2.0 + (double)x0.B * Wout * Hout * N * 4.0;
            Plan* pl = plan;
            plan->ops.push_back({[p, pl](hipStream_t s) mutable {
                p.y_nchw = pl->io.out;
                p.sch = pl->io.sch;
                return launch_conv_o4(p, s);
            }, "conv_o4_kernel<128,32,taps9>", fl, by});
        }
        *out = Tensor();
        return 0;
    }

    // conv_regw.hip, conv_ds2_kernel (round 4): stride-2 convs of 256 raw channels onto few pixels (the 64x4 -> 32x2 down-sampler); RLDM_NO_DS2=1 keeps
    // the generic kernel
    static bool ds2_params(const ConvArgs& a, int Cin_t, int R_t, int taps, int Wout, int Hout, ConvParams* q) {
        static const bool off = getenv("RLDM_NO_DS2") != nullptr;
        if (off || (dbg() & 2048) || g_force_bm || taps != 9 || a.stride != 2 || a.pad_mode != 0 || a.up != 1 || a.out_f32_nchw) return false;
        if (a.x1.valid() || a.gn || a.temb_off >= 0 || R_t != 0 || Cin_t != 256 || a.layer->Cin != 256 || a.layer->Cout % 32 != 0 || Wout % 32 != 0) return false;
        memset(q, 0, sizeof(*q));
        q->C0 = 256;
        q->B = a.x0.B; q->Win = a.x0.W; q->Hin = a.x0.H;
        q->up = 1; q->stride = 2; q->pad_lo = 1;
        q->Wout = Wout; q->Hout = Hout;
        q->TW = 32; q->TH = 1;
        q->tiles_h = Hout;
        q->tiles_img = (Wout / 32) * Hout;
        q->N = a.layer->Cout;
        q->ntile_n = q->N / 32;
        q->ksplit = 1;
        const long long blocks = (long long)q->B * q->tiles_img * q->ntile_n;
        return conv_ds2_supported(*q) && blocks >= 32 && blocks <= 512;      // (more: conv_stream's stride-2 instance or the generic kernel fill the chip)
    }
    int conv_ds2(const ConvArgs& a, int Wout, int Hout, Tensor* out) {
        ConvLayer* L = a.layer;
        const int N = L->Cout;
        const Tensor& x0 = a.x0;
        ConvParams p;
        RLDM_REQUIRE(ds2_params(a, x0.C, 0, 9, Wout, Hout, &p), "conv " + L->name + ": conv_ds2 route lost");
        Tensor y = make(x0.B, Wout, Hout, N);
        if (a.want_stats) add_stats(y, p.tiles_img);
        const double fl = 2.0 * (double)x0.B * Wout * Hout * N * (double)L->Cin * 9;
        plan->flops += fl;
        note_launch();
        if (!dry) {
            ConvLayer::Packed* pk = nullptr;
            if (L->get_streampacked(256, 2, &pk)) return 1;
            p.x0 = tptr(x0);
            p.wpk = pk->w.as<bf16_t>();
            p.bias = pk->bias.as<float>();
            p.y = tptr(y);
            p.y_ld = N;
            p.y_stats = y.P ? ptr<float2>(y.st_off) : nullptr;
            const double by = (double)x0.B * x0.W * x0.H * 256 * 2.0 + (double)N * L->Cin * 9 * 2.0 + (double)x0.B * Wout * Hout * N * 2.0;
            plan->ops.push_back({[p](hipStream_t s) { return launch_conv_ds2(p, s); }, "conv_ds2_kernel<32,32,taps9,s2>", fl, by});
        }
        *out = y;
        return 0;
    }

    // conv_regw.hip route (round 4): 64 -> 64 channel 3x3 convs over many 16 x 8 tiles (the VAE decoder's full-resolution level) -- the weights stay
    // in registers, a workgroup walks a run of tiles; rldm_debug_set_flags2(1 << 24) keeps them on conv_stream's per-tile instance
    static bool regw_params(const ConvArgs& a, int Cin_t, int R_t, int taps, int Wout, int Hout, ConvParams* q) {
        const int N_ = a.layer->Cout;
        if ((dbg2() & (1 << 24)) || (dbg() & 2048) || g_force_bm || taps != 9 || a.stride != 1 || a.pad_mode != 0 || a.up != 1) return false;
        if (a.x1.valid() || a.temb_off >= 0 || Cin_t != 64 || a.layer->Cin != 64 || a.first_of_step) return false;
        if (a.out_f32_nchw ? (N_ > 4 || R_t != 0 || getenv("RLDM_NO_RW_OUT") != nullptr) : N_ != 64) return false;
        if (R_t != 0 && !(a.layer->sc_identity && R_t == 64 && a.r0.valid() && a.r0.C == 64)) return false;
        if (Wout % 16 != 0 || Hout % 8 != 0) return false;
        memset(q, 0, sizeof(*q));
        q->C0 = 64; q->R0 = R_t;
        q->B = a.x0.B; q->Win = a.x0.W; q->Hin = a.x0.H;
        q->up = 1; q->stride = 1; q->pad_lo = 1;
        q->Wout = Wout; q->Hout = Hout;
        q->TW = 16; q->TH = 8; q->th_shift = 3;
        ConvTile t;
        t.BM = 256; t.BN = 128; t.CK = 64; t.taps = 9;
        q->colb = conv_halo_col_bytes(t, 8, 1);
        q->tiles_h = Hout / 8;
        q->tiles_img = (Wout / 16) * q->tiles_h;
        q->magic_thv = ((1 << 20) + 9) / 10;
        const int cpg = std::max(1, Cin_t / a.groups);
        q->magic_cpg = ((1 << 20) + cpg - 1) / cpg;
        q->gn_inv_n = (float)(1.0 / ((double)a.x0.W * a.x0.H * cpg));
        q->N = N_;
        q->silu = a.silu;
        q->gn_eps = a.eps;
        q->gn_groups = a.groups;
        q->ksplit = 1;
        q->exp = ((dbg2() & (1 << 25)) ? 8 : 0) | (getenv("RLDM_RW_ABL") ? atoi(getenv("RLDM_RW_ABL")) << 16 : 0);   // (tests: at most 8 team runs, so that small images give runs of several tiles; tuning)
        if (a.gn) q->st0 = reinterpret_cast<const float2*>(q);      // (only their presence matters to the shape check)
        if (a.out_f32_nchw) q->y_nchw = reinterpret_cast<float*>(q);
        const bool ok = conv_regw_supported(*q);
        q->st0 = nullptr;
        q->y_nchw = nullptr;
        return ok;
    }
    int conv_regw(const ConvArgs& a, int Cin_t, int R_t, int Wout, int Hout, Tensor* out) {
        ConvLayer* L = a.layer;
        const int N = L->Cout;
        const Tensor& x0 = a.x0;
        ConvParams p;
        RLDM_REQUIRE(regw_params(a, Cin_t, R_t, 9, Wout, Hout, &p), "conv " + L->name + ": conv_regw route lost");
        if (a.gn) {
            RLDM_REQUIRE(a.gn->C == Cin_t && Cin_t % a.groups == 0, "conv " + L->name + ": GroupNorm channel mismatch");
            RLDM_REQUIRE(x0.P > 0, "conv " + L->name + ": GroupNorm input without statistics");
        }
        p.dbg = dbg();
        p.ts = g_ts_buf;
        p.ntile_n = 1;
        Tensor y;
        if (!a.out_f32_nchw) {
            y = make(x0.B, Wout, Hout, N);
            if (a.want_stats) add_stats(y, conv_regw_partials(p));
        }
        const double fl = 2.0 * (double)x0.B * Wout * Hout * N * (double)L->Cin * 9;
        plan->flops += fl;
        note_launch();
        if (!dry) {
            ConvLayer::Packed* pk = nullptr;
            if (L->get_streampacked(Cin_t, 1, &pk)) return 1;
            p.x0 = tptr(x0);
            p.r0 = tptr(a.r0);
            p.wpk = pk->w.as<bf16_t>();
            p.bias = pk->bias.as<float>();
            if (a.gn) {
                p.st0 = sptr(x0);
                p.P0 = x0.P;
                p.gn_gamma = a.gn->gamma.as<float>();
                p.gn_beta = a.gn->beta.as<float>();
            }
            if (y.valid()) {
                p.y = tptr(y);
                p.y_ld = N;
                p.y_stats = y.P ? ptr<float2>(y.st_off) : nullptr;
            }
            const double by = (double)x0.B * x0.W * x0.H * Cin_t * 2.0 + (double)N * L->Cin * 9 * 2.0 +
                              (double)x0.B * Wout * Hout * N * (a.out_f32_nchw ? 4.0 : 2.0) + (double)x0.B * Wout * Hout * R_t * 2.0;
            Plan* pl = plan;
            const bool f32out = a.out_f32_nchw;
            plan->ops.push_back({[p, pl, f32out](hipStream_t s) mutable {
                if (f32out) {
                    RLDM_REQUIRE(pl->io.sch.coef_table == nullptr, "conv_regw: a scheduler step fused into this output layer");
                    p.y_nchw = pl->io.out;
                }
                return launch_conv_regw(p, s);
            }, std::string("conv_regw_kernel<128,") + (f32out ? "32" : "64") + ",taps9>", fl, by});
        }
        *out = y;
        return 0;
    }

    // Statistics of a tensor with many pixel tiles per image are folded once, by one small launch, instead of by every
    // workgroup of every consumer (gn_fold_kernel); RLDM_DBG_FLAGS=262144 keeps the raw partials for A/B runs.
    static inline const int kFoldAboveP = getenv("RLDM_FOLD_ABOVE") ? atoi(getenv("RLDM_FOLD_ABOVE")) : 32;   // (env: tuning runs)
    int fold_stats(Tensor& y) {
        if (!y.valid() || y.P <= kFoldAboveP || (dbg() & 262144)) return 0;
        const size_t raw_off = y.st_off, raw_bytes = y.st_bytes();
        const int rawP = y.P;
        add_stats(y, 2);
        note_launch();
        if (!dry) {
            GnFoldParams g;
            g.part = ptr<float2>(raw_off);
            g.out = ptr<float2>(y.st_off);
            g.B = y.B; g.P = rawP; g.C = y.C;
            plan->ops.push_back({[g](hipStream_t s) { return launch_gn_fold(g, s); }, "gn_fold_kernel", 0.0, (double)raw_bytes});
        }
        arena.release(raw_off, raw_bytes);
        return 0;
    }

    // y = conv(...) ; consumes nothing (callers release inputs)
    int conv(const ConvArgs& a0, Tensor* out) {
        ConvArgs a = a0;
        const int ord = conv_ord++;
        ConvLayer* L = a.layer;
        RLDM_REQUIRE(L != nullptr, "internal: missing conv layer");
        const int taps = L->ksize * L->ksize;
        const int Cin_t = a.x0.C + (a.x1.valid() ? a.x1.C : 0);
        const int R_t = (a.r0.valid() ? a.r0.C : 0) + (a.r1.valid() ? a.r1.C : 0);
        const int Wout = a.x0.W * a.up / a.stride, Hout = a.x0.H * a.up / a.stride;
        Tensor view;
        if (recording && vp) {
            // as a consumer: could this conv read cat[x0, x1] already normalised + activated?  (3x3 on the conv_small route)
            ConvArgs c = a;
            c.x0.C = Cin_t; c.x1 = Tensor(); c.gn = nullptr; c.silu = 0;
            if (a.gn) record_consumer(a.gn, a.x0, a.x1, a.silu, taps == 9 && small_route(c, Cin_t, R_t, taps, Wout, Hout), a.groups, a.eps);
            // as a producer: with one tile per image, whether its own input arrives raw (statistics) or pre-activated
            ConvArgs o = a, oc = c;
            o.own_image = oc.own_image = true;
            if ((int)vp->can_emit.size() <= ord) vp->can_emit.resize(ord + 1, 0);
            if ((int)vp->out_c.size() <= ord) vp->out_c.resize(ord + 1, 0);
            vp->out_c[ord] = L->Cout;
            vp->can_emit[ord] = !(dbg() & 1048576) && !a.out_f32_nchw && small_route(o, Cin_t, R_t, taps, Wout, Hout) &&
                                (!a.gn || small_route(oc, Cin_t, R_t, taps, Wout, Hout));
        } else if (vp) {
            if (take_view(a.gn, &view)) {
                a.x0 = view; a.x1 = Tensor(); a.gn = nullptr; a.silu = 0;
            }
            auto it = vp->emit.find(ord);
            cur_emit = (it != vp->emit.end() && !it->second.empty()) ? &it->second : nullptr;
            // one tile per image: to normalise for the consumers, and (input ready, nothing to fold) to be a trunk phase
            a.own_image = cur_emit != nullptr ||
                          (trunk_tiles() && !a.gn && ord < (int)vp->can_emit.size() && vp->can_emit[ord] && !a.x1.valid());
        }
        const size_t ops_before = plan->ops.size(), ph_before = pend.standalone.size();
        if (conv_route(a, out)) return 1;
        if (!dry) {
            for (size_t i = ops_before; i < plan->ops.size(); ++i) plan->ops[i].tag += L->name + " ";
            for (size_t i = ph_before; i < pend.standalone.size(); ++i) pend.standalone[i].tag += L->name + " ";
        }
        cur_emit = nullptr;
        if (view.valid()) release(view);
        out->prod = ord;
        if (out->valid()) live[out->id] = *out;
        return fold_stats(*out);
    }
    int conv_route(const ConvArgs& a, Tensor* out) {
        ConvLayer* L = a.layer;
        RLDM_REQUIRE(L != nullptr, "internal: missing conv layer");
        const Tensor& x0 = a.x0;
        const int Cin_t = x0.C + (a.x1.valid() ? a.x1.C : 0);      // tensor (padded) channels
        RLDM_REQUIRE(Cin_t >= L->Cin, "conv " + L->name + ": input tensor has fewer channels than the weights");
        RLDM_REQUIRE(Cin_t % 16 == 0, "conv " + L->name + ": input channels must be a multiple of 16");
        const int R_t = (a.r0.valid() ? a.r0.C : 0) + (a.r1.valid() ? a.r1.C : 0);
        RLDM_REQUIRE(R_t == L->R, "conv " + L->name + ": residual sources do not match the layer's residual phase");
        const int taps = L->ksize * L->ksize;
        const int Wv = x0.W * a.up, Hv = x0.H * a.up;
        const int Wout = Wv / a.stride, Hout = Hv / a.stride;
        RLDM_REQUIRE(Wv % a.stride == 0 && Hv % a.stride == 0, "conv " + L->name + ": odd size under stride 2");
        RLDM_REQUIRE(R_t == 0 || (a.r0.W == Wout && a.r0.H == Hout), "conv " + L->name + ": residual resolution mismatch");
        const int N = L->Cout;
        // (conv_in stays on the generic kernel: it is the launch that advances the sampler's step index)
        if (!a.first_of_step && small_route(a, Cin_t, R_t, taps, Wout, Hout)) return conv_small(a, Cin_t, R_t, taps, Wout, Hout, out);
        {
            ConvParams q;
            if (c16_params(a, Cin_t, R_t, taps, Wout, Hout, &q)) return conv_c16(a, Wout, Hout, out);
            if (o4_params(a, Cin_t, R_t, taps, Wout, Hout, &q)) return conv_o4(a, Wout, Hout, out);
            if (!a.first_of_step && ds2_params(a, Cin_t, R_t, taps, Wout, Hout, &q)) return conv_ds2(a, Wout, Hout, out);
            if (regw_params(a, Cin_t, R_t, taps, Wout, Hout, &q)) return conv_regw(a, Cin_t, R_t, Wout, Hout, out);
        }
        if (!a.first_of_step) {
            ConvParams q;
            if (stream_params(a, Cin_t, R_t, taps, Wout, Hout, &q)) return conv_stream(a, Cin_t, R_t, Wout, Hout, out);
        }
        const TileChoice tc = choose_tile(x0.B, Wout, Hout, a.stride, N, Cin_t, x0.C, R_t, a.r0.valid() ? a.r0.C : 0, taps,
                                          a.out_f32_nchw, a.gn != nullptr);
        const ConvTile tile = tc.tile;
        RLDM_REQUIRE(conv_tile_supported(tile), "conv " + L->name + ": no kernel instance");

        ConvParams p;
        memset(&p, 0, sizeof(p));
        p.C0 = x0.C;
        p.C1 = a.x1.valid() ? a.x1.C : 0;
        p.R0 = a.r0.valid() ? a.r0.C : 0;
        p.R1 = a.r1.valid() ? a.r1.C : 0;
        p.B = x0.B; p.Win = x0.W; p.Hin = x0.H;
        p.up = a.up; p.stride = a.stride;
        p.pad_lo = (L->ksize == 1) ? 0 : (a.pad_mode == 0 ? 1 : 0);
        p.Wout = Wout; p.Hout = Hout;
        p.TW = tc.TW; p.TH = tc.TH;
        p.colb = conv_halo_col_bytes(tile, tc.TH, a.stride);
        p.tiles_h = Hout / p.TH;
        p.tiles_img = (Wout / p.TW) * p.tiles_h;
        RLDM_REQUIRE((p.TH & (p.TH - 1)) == 0, "conv " + L->name + ": pixel tile height must be a power of two");
        p.th_shift = 0;
        while ((1 << p.th_shift) < p.TH) ++p.th_shift;
        {
            const int thv = (p.TH - 1) * a.stride + (L->ksize == 3 ? 3 : 1);
            p.magic_thv = ((1 << 20) + thv - 1) / thv;
            const int cpg = std::max(1, Cin_t / a.groups);
            p.magic_cpg = ((1 << 20) + cpg - 1) / cpg;
            p.gn_inv_n = (float)(1.0 / ((double)x0.W * x0.H * cpg));
        }
        RLDM_REQUIRE(Wout % p.TW == 0 && Hout % p.TH == 0, "conv " + L->name + ": size not tileable (powers of two expected)");
        p.N = N;
        p.silu = a.silu;
        p.gn_eps = a.eps;
        p.dbg = dbg();
        p.ts = (getenv("RLDM_TS_ATTN_L") || getenv("RLDM_TS_TRUNK")) ? nullptr : g_ts_buf;      // (the attention timeline owns the buffer then)
        if (getenv("RLDM_TS_ORD")) p.ts = atoi(getenv("RLDM_TS_ORD")) == conv_ord - 1 ? g_ts_buf : nullptr;
        p.gn_groups = a.groups;
        p.ksplit = tc.ksplit;
        const int tiles_img = (Wout / p.TW) * (Hout / p.TH);
        const int ntile_n = (N + tile.BN - 1) / tile.BN;
        const long long tiles = (long long)x0.B * tiles_img * ntile_n;

        Tensor y;
        if (!a.out_f32_nchw) {
            y = make(x0.B, Wout, Hout, N);
            if (a.want_stats) add_stats(y, tiles_img);
        }
        if (a.gn) {
            RLDM_REQUIRE(a.gn->C == Cin_t && Cin_t % a.groups == 0, "conv " + L->name + ": GroupNorm channel mismatch");
            RLDM_REQUIRE(x0.P > 0 && (!a.x1.valid() || a.x1.P > 0), "conv " + L->name + ": GroupNorm input without statistics");
        }
        size_t slab_off = 0, slab_bytes = 0;
        int ticket_base = 0;
        if (tc.ksplit > 1) {
            slab_bytes = (size_t)tiles * tc.ksplit * tile.BM * tile.BN * sizeof(float);
            slab_off = arena.alloc(slab_bytes);
            ticket_base = tickets;
            tickets += (int)tiles;
        }
        plan->flops += 2.0 * (double)x0.B * Wout * Hout * N * ((double)L->Cin * taps + (L->sc_identity ? 0.0 : (double)L->R));
        note_launch();
        if (!dry) {
            ConvLayer::Packed* pk = nullptr;
            if (L->get_packed(tile.BN, tile.CK, Cin_t, &pk)) return 1;
            p.x0 = tptr(x0);
            p.x1 = tptr(a.x1);
            p.r0 = tptr(a.r0);
            p.r1 = tptr(a.r1);
            p.wpk = pk->w.as<bf16_t>();
            p.bias = pk->bias.as<float>();
            p.ntile_n = pk->ntile_n;
            if (a.gn) {
                p.st0 = sptr(x0);
                p.st1 = sptr(a.x1);
                p.P0 = x0.P;
                p.P1 = a.x1.valid() ? a.x1.P : 0;
                p.gn_gamma = a.gn->gamma.as<float>();
                p.gn_beta = a.gn->beta.as<float>();
            }
            p.y = tptr(y);
            p.y_ld = N;
            p.y_stats = (y.valid() && y.P) ? ptr<float2>(y.st_off) : nullptr;
            if (tc.ksplit > 1) {
                p.slab = ptr<float>(slab_off);
                p.ticket = ticket_ptr + ticket_base;
            }
            Plan* pl = plan;
            p.temb_ld = temb_ld;
            const int temb_off = a.temb_off;
            const bool f32out = a.out_f32_nchw;
            const bool first_of_step = a.first_of_step;
            const double fl = 2.0 * (double)x0.B * Wout * Hout * N * ((double)L->Cin * taps + (L->sc_identity ? 0.0 : (double)L->R));
            const double by = (double)x0.B * x0.W * x0.H * Cin_t * 2.0 + (double)N * (L->Cin * taps + L->R) * 2.0 +
                              (double)x0.B * Wout * Hout * N * (a.out_f32_nchw ? 4.0 : 2.0) +
                              (double)x0.B * Wout * Hout * R_t * 2.0;
            const std::string kname = "conv_igemm_kernel<" + std::to_string(tile.BM) + "," + std::to_string(tile.BN) +
                                      ",CK" + std::to_string(tile.CK) + ",taps" + std::to_string(tile.taps) + ">";
            plan->ops.push_back({[p, tile, pl, temb_off, f32out, first_of_step](hipStream_t s) mutable {
                p.step_inc = (first_of_step && pl->io.pack_fused) ? pl->io.step_inc : nullptr;
                if (temb_off >= 0) {
                    p.temb = pl->io.temb + temb_off;
                    p.step_ptr = pl->io.step_ptr;
                    p.temb_rows_per_step = pl->io.temb_rows_per_step;
                    p.temb_per_sample = pl->io.temb_per_sample;
                }
                if (f32out) {
                    p.y_nchw = pl->io.out;
                    p.sch = pl->io.sch;
                }
                return launch_conv(tile, p, s);
            }, kname, fl, by});
        }
        if (tc.ksplit > 1) arena.release(slab_off, slab_bytes);
        *out = y;
        return 0;
    }
};

// Two passes over the same walk: a dry one sizes the activation arena (and the split-K counters), the real one bakes
// device pointers into the launch list.
struct PlanFlagScope {            // t_plan_flags = the plan's options for the duration of its construction
    int saved;
    explicit PlanFlagScope(int f) : saved(t_plan_flags) { t_plan_flags = f; }
    ~PlanFlagScope() { t_plan_flags = saved; }
};

static int build_plan(Plan* plan, int temb_ld, const std::function<int(Builder&)>& walk, int* launches = nullptr) {
    PlanFlagScope scope(plan->flags);
    // pass 0 records which GroupNorms could be applied by the producers of their inputs (ViewPlan)
    ViewPlan vplan;
    {
        Builder rec;
        rec.plan = plan;
        rec.dry = true;
        rec.recording = true;
        rec.vp = &vplan;
        rec.temb_ld = temb_ld;
        if (walk(rec)) return 1;
        vplan.decide();
        // tuning aid (tools/bench_conv.py on a single conv): RLDM_FAKE_VIEWS=n makes every conv that could normalise for a
        // consumer write n copies nobody reads (identity affine, SiLU on), so the epilogue's cost can be stamped alone
        const char* fv = getenv("RLDM_FAKE_VIEWS");
        if (!fv && (dbg() & (1 << 22))) fv = "2";     // (tests: the own-image tile + epilogue on single-conv plans)
        if (fv) {
            static std::map<int, std::unique_ptr<NormParams>> dummies;
            for (int ord = 0; ord < (int)vplan.can_emit.size(); ++ord) {
                if (!vplan.can_emit[ord] || vplan.emit.count(ord) || vplan.out_c[ord] % 32 != 0) continue;
                const int C = vplan.out_c[ord];
                for (int i = 0; i < std::min(3, atoi(fv)); ++i) {
                    auto& d = dummies[C * 4 + i];
                    if (!d) {
                        d = std::make_unique<NormParams>();
                        d->C = C;
                        std::vector<float> one(C, 1.f), zero(C, 0.f);
                        if (upload(d->gamma, one.data(), C * 4) || upload(d->beta, zero.data(), C * 4)) return 1;
                    }
                    vplan.emit[ord].push_back({d.get(), 0, C, 1, 32, 1e-5f});
                }
            }
        }
        plan->flops = 0;
    }
    Builder dry;
    dry.plan = plan;
    dry.dry = true;
    dry.temb_ld = temb_ld;
    dry.vp = &vplan;
    if (walk(dry)) return 1;
    if (dry.flush_trunk()) return 1;
    if (launches) *launches = dry.launches;
    if (plan->arena.alloc(dry.arena.peak + 256)) return 1;
    if (plan->tickets.alloc((size_t)std::max(1, dry.tickets) * sizeof(int))) return 1;
    RLDM_HIP_CHECK(hipMemset(plan->tickets.p, 0, plan->tickets.bytes));
    plan->flops = 0;
    plan->ops.clear();
    Builder real;
    real.plan = plan;
    real.dry = false;
    real.base = plan->arena.as<char>();
    real.ticket_ptr = plan->tickets.as<int>();
    real.temb_ld = temb_ld;
    real.vp = &vplan;
    if (walk(real)) return 1;
    if (real.flush_trunk()) return 1;
    if (getenv("RLDM_PRINT_PLAN")) {                 // tuning aid: the launch list of every plan built
        fprintf(stderr, "plan B=%d %dx%d: %zu launches\n", plan->B, plan->W, plan->H, plan->ops.size());
        for (size_t i = 0; i < plan->ops.size(); ++i)
            fprintf(stderr, "  %3zu %-64s %8.3f GFLOP %8.2f MB  %s\n", i, plan->ops[i].name.c_str(), plan->ops[i].flops * 1e-9,
                    plan->ops[i].bytes * 1e-6, plan->ops[i].tag.c_str());
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// shared network pieces
// ---------------------------------------------------------------------------------------------------------------
struct NetCommon {
    Layers layers;
    int groups = 32;
    float eps = 1e-5f;
    int temb_ld = 0;                                 // total projected channels (0: no time embedding)
    std::map<std::string, int> temb_off;             // resnet prefix -> channel offset in the temb row

    // ---- a concatenation wider than the low-resolution conv kernel takes (RangeDM's 512-channel levels: 512 + 512 and 512 + 256
    // input channels, ldm/configs/RangeDM.yaml:19-21) -----------------------------------------------------------------------
    // conv_small.hip copies ALL input channels of its pixel tile to LDS (<= 512); past that the generic kernel walked K = 9216 on a
    // handful of workgroups (54-63 us per conv, 36 % of a RangeDM step at 0.9 % of the MFMA peak).  The convolution is linear in
    // its input channels, so the block runs as
    //     (a0, a1) = silu(GN1(cat[x, skip]))            one gn_apply launch, two output tensors (x.C | skip.C channels)
    //     t  = conv1[:, :x.C](a0)                       no bias
    //     h1 = conv1[:, x.C:](a1) + b1 + temb + t       t enters as the epilogue's identity residual
    //     u  = conv2(silu(GN2(h1))) + shortcut[:, :x.C](x) + b2 + b_sc       (the K-phase residual over x alone)
    //     y  = shortcut[:, x.C:](skip) + u              a 1x1 conv with u as its identity residual
    // -- four launches of the fast kernel for two of the slow one; t and u are rounded to bf16 on the way (one more rounding of a
    // partial sum: inside the network tolerance, tests/test_hip_models.py::test_other_presets_full_size_match_reference).
    // rldm_debug_set_flags(1 << 21) keeps the two generic launches (A/B runs).
    bool wide_concat(const Builder& b, const Tensor& x, const Tensor& skip) const {
        if (dbg() & (1 << 21)) return false;
        if (!skip.valid() || x.C + skip.C <= 512 || x.C > 512 || skip.C > 512 || x.C % 64 != 0 || skip.C % 64 != 0) return false;
        const long long px = (long long)x.W * x.H;
        (void)b;
        return (px <= 256 || (long long)x.B * px <= 4096) && x.H >= 2;      // where conv_small.hip would run the halves
    }
    ConvLayer* derived_conv(const std::string& name, const ConvLayer* src, int c_lo, int c_hi, bool keep_bias) {
        if (ConvLayer* L = layers.get_conv(name)) return L;
        auto L = std::make_unique<ConvLayer>();
        const int taps = src->ksize * src->ksize, Ci = c_hi - c_lo;
        L->name = name;
        L->Cout = src->Cout; L->Cin = Ci; L->ksize = src->ksize;
        L->w.resize((size_t)src->Cout * Ci * taps);
        for (int n = 0; n < src->Cout; ++n)
            for (int c = 0; c < Ci; ++c)
                for (int t = 0; t < taps; ++t)
                    L->w[((size_t)n * Ci + c) * taps + t] = src->w[((size_t)n * src->Cin + c_lo + c) * taps + t];
        L->b = keep_bias ? src->b : std::vector<float>(src->Cout, 0.f);
        ConvLayer* raw = L.get();
        layers.conv[name] = std::move(L);
        return raw;
    }
    int resnet_wide(Builder& b, const std::string& p, Tensor x, Tensor skip, Tensor* out) {
        ConvLayer* c1 = layers.get_conv(p + ".conv1");
        ConvLayer* c2 = layers.get_conv(p + ".conv2");
        NormParams* n1 = layers.get_norm(p + ".norm1");
        const int Ctot = x.C + skip.C;
        RLDM_REQUIRE(c1 && c2 && n1 && c1->Cin == Ctot && c2->R == Ctot && !c2->sc_identity, "resnet " + p + ": not a wide concatenation");
        // -- GroupNorm + SiLU of the concatenation, once, into the two halves
        Tensor a0 = b.make(x.B, x.W, x.H, x.C), a1 = b.make(x.B, x.W, x.H, skip.C);
        RLDM_REQUIRE(x.P > 0 && skip.P > 0 && Ctot % groups == 0, "resnet " + p + ": GroupNorm input without statistics");
        b.note_launch();
        if (!b.dry) {
            GnApplyParams g;
            memset(&g, 0, sizeof(g));
            g.x0 = b.tptr(x); g.x1 = b.tptr(skip);
            g.C0 = x.C; g.C1 = skip.C;
            g.st0 = b.sptr(x); g.st1 = b.sptr(skip);
            g.P0 = x.P; g.P1 = skip.P;
            g.B = x.B; g.npix = x.W * x.H;
            g.groups = groups;
            g.gamma = n1->gamma.as<float>(); g.beta = n1->beta.as<float>();
            g.eps = eps; g.silu = 1;
            g.y = b.tptr(a0); g.y1 = b.tptr(a1); g.ysplit = x.C;
            b.plan->ops.push_back({[g](hipStream_t s) { return launch_gn_apply(g, s); }, "gn_apply_kernel", 0.0,
                                   (double)g.B * g.npix * Ctot * 4.0});
        }
        // -- conv1 in two halves
        ConvLayer* c1a = derived_conv(p + ".conv1@lo", c1, 0, x.C, false);
        ConvLayer* c1b = derived_conv(p + ".conv1@hi", c1, x.C, Ctot, true);
        c1b->R = c1b->Cout;
        c1b->sc_identity = true;
        ConvArgs A;
        A.layer = c1a; A.x0 = a0;
        Tensor t;
        if (b.conv(A, &t)) return 1;
        b.release(a0);
        ConvArgs B_;
        B_.layer = c1b; B_.x0 = a1; B_.r0 = t;
        auto it = temb_off.find(p);
        B_.temb_off = it == temb_off.end() ? -1 : it->second;
        B_.want_stats = true;
        Tensor h1;
        if (b.conv(B_, &h1)) return 1;
        b.release(a1);
        b.release(t);
        // -- conv2 with the shortcut over x in its K loop, then the shortcut over skip as a pointwise conv on top
        ConvLayer* c2a = layers.get_conv(p + ".conv2@lo");
        if (!c2a) {
            auto L = std::make_unique<ConvLayer>();         // (own host weights; the packed images are built per layer)
            L->name = p + ".conv2@lo";
            L->Cout = c2->Cout; L->Cin = c2->Cin; L->ksize = c2->ksize;
            L->w = c2->w;
            L->b = c2->b;
            L->R = x.C;
            L->sc_w.resize((size_t)c2->Cout * x.C);
            for (int n = 0; n < c2->Cout; ++n)
                for (int c = 0; c < x.C; ++c) L->sc_w[(size_t)n * x.C + c] = c2->sc_w[(size_t)n * Ctot + c];
            c2a = L.get();
            layers.conv[L->name] = std::move(L);
        }
        ConvLayer* c2b = layers.get_conv(p + ".conv2@hi");
        if (!c2b) {
            auto L = std::make_unique<ConvLayer>();
            L->name = p + ".conv2@hi";
            L->Cout = c2->Cout; L->Cin = skip.C; L->ksize = 1;
            L->w.resize((size_t)c2->Cout * skip.C);
            for (int n = 0; n < c2->Cout; ++n)
                for (int c = 0; c < skip.C; ++c) L->w[(size_t)n * skip.C + c] = c2->sc_w[(size_t)n * Ctot + x.C + c];
            L->b.assign(c2->Cout, 0.f);
            L->R = c2->Cout;
            L->sc_identity = true;
            c2b = L.get();
            layers.conv[L->name] = std::move(L);
        }
        ConvArgs C2;
        C2.layer = c2a; C2.x0 = h1;
        C2.gn = layers.get_norm(p + ".norm2");
        C2.eps = eps; C2.groups = groups; C2.silu = 1;
        C2.r0 = x;
        Tensor u;
        if (!b.small_route(C2, h1.C, x.C, 9, h1.W, h1.H)) {
            // the 3x3 tile and the shortcut's input tile do not fit the LDS together (64x4 images: 112 + 67 KB): the shortcut over x
            // becomes a pointwise conv of its own as well --  u0 = conv2(..) + b;  u = shortcut[:, :x.C](x) + u0
            ConvLayer* c2m = layers.get_conv(p + ".conv2@main");
            if (!c2m) {
                auto L = std::make_unique<ConvLayer>();
                L->name = p + ".conv2@main";
                L->Cout = c2->Cout; L->Cin = c2->Cin; L->ksize = c2->ksize;
                L->w = c2->w;
                L->b = c2->b;
                c2m = L.get();
                layers.conv[L->name] = std::move(L);
            }
            ConvLayer* c2x = layers.get_conv(p + ".conv2@x");
            if (!c2x) {
                auto L = std::make_unique<ConvLayer>();
                L->name = p + ".conv2@x";
                L->Cout = c2->Cout; L->Cin = x.C; L->ksize = 1;
                L->w = c2a->sc_w;
                L->b.assign(c2->Cout, 0.f);
                L->R = c2->Cout;
                L->sc_identity = true;
                c2x = L.get();
                layers.conv[L->name] = std::move(L);
            }
            C2.layer = c2m;
            C2.r0 = Tensor();
            Tensor u0;
            if (b.conv(C2, &u0)) return 1;
            ConvArgs X;
            X.layer = c2x; X.x0 = x; X.r0 = u0;
            if (b.conv(X, &u)) return 1;
            b.release(u0);
        } else if (b.conv(C2, &u)) {
            return 1;
        }
        b.release(h1);
        ConvArgs D;
        D.layer = c2b; D.x0 = skip; D.r0 = u;
        D.want_stats = true;
        if (b.conv(D, out)) return 1;
        b.release(u);
        b.release(x);
        b.release(skip);
        return 0;
    }

    // ResnetBlock2D / sgm ResnetBlock (model.py:342-362) in two launches: conv1 = GN1+SiLU+conv+temb, conv2 =
    // GN2+SiLU+conv with the shortcut (1x1 conv or identity) as the residual K-phase.  Releases one reference of x and skip.
    int resnet(Builder& b, const std::string& p, Tensor x, Tensor skip, Tensor* out) {
        if (wide_concat(b, x, skip)) return resnet_wide(b, p, x, skip, out);
        ConvArgs c1;
        c1.layer = layers.get_conv(p + ".conv1");
        c1.x0 = x; c1.x1 = skip;
        c1.gn = layers.get_norm(p + ".norm1");
        c1.eps = eps; c1.groups = groups; c1.silu = 1;
        auto it = temb_off.find(p);
        c1.temb_off = it == temb_off.end() ? -1 : it->second;
        c1.want_stats = true;
        Tensor h1;
        if (b.conv(c1, &h1)) return 1;
        ConvArgs c2;
        c2.layer = layers.get_conv(p + ".conv2");
        c2.x0 = h1;
        c2.gn = layers.get_norm(p + ".norm2");
        c2.eps = eps; c2.groups = groups; c2.silu = 1;
        c2.r0 = x; c2.r1 = skip;
        c2.want_stats = true;
        if (b.conv(c2, out)) return 1;
        b.release(h1);
        b.release(x);
        b.release(skip);
        return 0;
    }

    // diffusers Attention (+x residual): qkv = GN + Linear(C, 3C); softmax(q k^T / sqrt(8)) v per head; to_out + x.
    // Releases one reference of x.
    // per-head q/k/v fragments for attention_qkv_d8_kernel, built once per layer from the merged qkv Linear
    struct AttnFused {
        DevBuf w, bias;
    };
    std::map<std::string, std::unique_ptr<AttnFused>> attn_fused;
    int get_attn_fused(const std::string& p, int C, AttnFused** out) {
        auto it = attn_fused.find(p);
        if (it != attn_fused.end()) {
            *out = it->second.get();
            return 0;
        }
        ConvLayer* L = layers.get_conv(p + ".qkv");
        RLDM_REQUIRE(L && L->Cin == C && L->Cout == 3 * C, "attention " + p + ": unexpected q/k/v shapes");
        std::vector<bf16_t> img;
        std::vector<float> bias;
        pack_attn_head_frags(L->w.data(), L->b.data(), C, img, bias);
        auto f = std::make_unique<AttnFused>();
        if (upload(f->w, img.data(), img.size() * sizeof(bf16_t))) return 1;
        if (upload(f->bias, bias.data(), bias.size() * sizeof(float))) return 1;
        *out = f.get();
        attn_fused[p] = std::move(f);
        return 0;
    }

    int attention(Builder& b, const std::string& p, Tensor x, Tensor* out) {
        const int Lt = x.W * x.H;
        if (!(dbg() & 32768) && x.C % 16 == 0 && x.C <= 512 && Lt <= 1024 && x.P > 0 && x.C % groups == 0) {
            // GroupNorm + q/k/v projection inside the attention launch: no [B][L][3C] tensor, one launch less
            NormParams* gnp = layers.get_norm(p + ".group_norm");
            b.record_consumer(gnp, x, Tensor(), 0, true, groups, eps);
            Tensor xn;                                  // x already normalised by its producer (producer-side GroupNorm)
            const bool pre = b.take_view(gnp, &xn);
            Tensor o = b.make(x.B, x.W, x.H, x.C);
            const double fl = 4.0 * (double)x.B * (x.C / 8) * (double)Lt * Lt * 8 + 2.0 * (double)x.B * Lt * 3.0 * x.C * x.C;
            b.plan->flops += fl;
            bool in_trunk = b.trunk_attention_ok(x, pre);           // a phase of the persistent trunk launch (trunk.hip)
            const int cl_ranks = in_trunk ? 0 : b.cluster_attention_ranks(x, pre);
            if (in_trunk) b.trunk_begin(x.B, x.C / Builder::own_tile_channels(x.B, Lt, x.C));
            else if (cl_ranks) b.trunk_begin(x.B, cl_ranks, x.C / 64, 2);
            else b.note_launch();
            in_trunk = in_trunk || cl_ranks != 0;
            // the stand-alone launch also carries the block's output projection (+ x, + statistics) when an image's workgroups are
            // 64-pixel blocks of it on one XCD, all resident: the 128x8 level at batch 8 / 16 (attention_body.h, attention_proj_tail);
            // rldm_debug_set_flags(128) keeps the projection a launch of its own
            // (without clusters -- a sampler's per-layer fall-back, several chains, the A/B switches -- the SAME tail runs as a launch of
            //  its own behind the attention launch: identical results)
            const bool proj_tail = !in_trunk && !(dbg() & 128) && attention_proj_fusable(x.B, Lt, x.C, -1);
            const bool proj_seam = proj_tail && Builder::cluster_enabled() &&
                                   attention_proj_fusable(x.B, Lt, x.C, Builder::device_cus() / std::max(1, g_concurrent_plans));
            const bool fuse_proj = proj_tail;
            Tensor yfused;
            if (fuse_proj) {
                yfused = b.make(x.B, x.W, x.H, x.C);
                b.add_stats(yfused, Lt / 64);
            }
            if (!b.dry) {
                AttnFused* f = nullptr;
                if (get_attn_fused(p, x.C, &f)) return 1;
                AttnQkvParams ap;
                memset(&ap, 0, sizeof(ap));
                ap.x = pre ? b.tptr(xn) : b.tptr(x);
                ap.st = pre ? nullptr : b.sptr(x);
                ap.P = x.P;
                ap.gamma = gnp->gamma.as<float>();
                ap.beta = gnp->beta.as<float>();
                ap.eps = eps;
                ap.groups = groups;
                const int cpg = x.C / groups;
                ap.inv_n = (float)(1.0 / ((double)Lt * cpg));
                ap.magic_cpg = ((1 << 20) + cpg - 1) / cpg;
                ap.wfrag = f->w.as<bf16_t>();
                ap.bias = f->bias.as<float>();
                ap.out = b.tptr(o);
                ap.B = x.B; ap.L = Lt; ap.C = x.C;
                ap.ts = (getenv("RLDM_TS_TRUNK") || getenv("RLDM_TS_ORD")) ? nullptr : g_ts_buf;
                ap.ts_L = getenv("RLDM_TS_ATTN_L") ? atoi(getenv("RLDM_TS_ATTN_L")) : 0;
                double by = (double)x.B * Lt * x.C * 2.0 * 2.0 + 3.0 * x.C * x.C * 2.0;
                double flp = fl;
                if (fuse_proj) {
                    ConvLayer* Lo = layers.get_conv(p + ".to_out.0");
                    ConvLayer::Packed* pk = nullptr;
                    RLDM_REQUIRE(Lo && Lo->Cin == x.C && Lo->Cout == x.C && Lo->sc_identity, "attention " + p + ": unexpected to_out");
                    if (Lo->get_fragpacked(x.C, 1, true, &pk)) return 1;
                    ap.proj_w = pk->w.as<bf16_t>();
                    ap.proj_bias = pk->bias.as<float>();
                    ap.proj_res = b.tptr(x);
                    ap.proj_y = b.tptr(yfused);
                    ap.proj_stats = b.ptr<float2>(yfused.st_off);
                    if (proj_seam) {
                        auto ctrs = std::make_unique<DevBuf>();
                        if (ctrs->alloc((size_t)x.B * 32 * 4)) return 1;
                        RLDM_HIP_CHECK(hipMemset(ctrs->p, 0, ctrs->bytes));
                        if (!b.plan->trunk_error.p) {
                            if (b.plan->trunk_error.alloc(64)) return 1;
                            RLDM_HIP_CHECK(hipMemset(b.plan->trunk_error.p, 0, 64));
                        }
                        ap.proj_counter = ctrs->as<unsigned>();
                        ap.proj_error = b.plan->trunk_error.as<int>();
                        b.plan->trunk_bufs.push_back(std::move(ctrs));
                        flp += 2.0 * (double)x.B * Lt * x.C * x.C;
                        by += (double)x.B * Lt * x.C * 2.0 * 3.0 + (double)x.C * x.C * 2.0;
                    }
                }
                Op standalone{[ap](hipStream_t s) { return launch_attention_qkv(ap, s); },
                              proj_seam ? "attention_qkv_d8_kernel + to_out" : "attention_qkv_d8_kernel", flp, by};
                standalone.tag = p;
                if (in_trunk) {
                    b.trunk_push_attention(ap, fl, by);
                    b.pend.standalone.push_back(standalone);
                } else {
                    b.plan->ops.push_back(standalone);
                }
                if (fuse_proj && !proj_seam) {          // the tail as its own launch (no seam: the kernel boundary orders it)
                    Op tail{[ap](hipStream_t s) { return launch_attention_proj(ap, s); }, "attention_proj_kernel",
                            2.0 * (double)x.B * Lt * x.C * x.C, (double)x.B * Lt * x.C * 2.0 * 3.0 + (double)x.C * x.C * 2.0};
                    tail.tag = p + ".to_out.0";
                    b.plan->ops.push_back(tail);
                }
            }
            if (fuse_proj && !proj_seam) b.note_launch();
            if (pre) b.release(xn);
            if (fuse_proj) {
                b.plan->flops += 2.0 * (double)x.B * Lt * x.C * x.C;
                b.release(o);
                b.release(x);
                *out = yfused;
                return 0;
            }
            ConvArgs co;
            co.layer = layers.get_conv(p + ".to_out.0");
            co.x0 = o;
            co.r0 = x;
            co.want_stats = true;
            if (b.conv(co, out)) return 1;
            b.release(o);
            b.release(x);
            return 0;
        }
        ConvArgs cq;
        cq.layer = layers.get_conv(p + ".qkv");
        cq.x0 = x;
        cq.gn = layers.get_norm(p + ".group_norm");
        cq.eps = eps; cq.groups = groups; cq.silu = 0;
        Tensor qkv;
        if (b.conv(cq, &qkv)) return 1;
        Tensor o = b.make(x.B, x.W, x.H, x.C);
        const int L = x.W * x.H;
        b.plan->flops += 4.0 * (double)x.B * (x.C / 8) * (double)L * L * 8;
        b.note_launch();
        if (!b.dry) {
            AttnParams ap;
            ap.qkv = b.tptr(qkv); ap.out = b.tptr(o);
            ap.B = x.B; ap.L = L; ap.C = x.C;
            b.plan->ops.push_back({[ap](hipStream_t s) { return launch_attention(ap, s); }, "attention_d8_kernel",
                                   4.0 * (double)x.B * (x.C / 8) * (double)L * L * 8, (double)x.B * L * x.C * 2.0 * 4.0});
        }
        b.release(qkv);
        ConvArgs co;
        co.layer = layers.get_conv(p + ".to_out.0");
        co.x0 = o;
        co.r0 = x;
        co.want_stats = true;
        if (b.conv(co, out)) return 1;
        b.release(o);
        b.release(x);
        return 0;
    }

    int add_resnet(ParamStore& ps, const std::string& p, int ci, int co, bool temb) {
        if (layers.add_norm(ps, p + ".norm1", ci)) return 1;
        if (layers.add_conv(ps, p + ".conv1", co, ci, 3)) return 1;
        if (layers.add_norm(ps, p + ".norm2", co)) return 1;
        if (layers.add_conv(ps, p + ".conv2", co, co, 3)) return 1;
        ConvLayer* c2 = layers.get_conv(p + ".conv2");
        if (ci != co) {                              // conv_shortcut(x) folded into conv2: weights appended, biases summed
            c2->R = ci;
            c2->sc_w = ps.host.at(p + ".conv_shortcut.weight");
            const auto& sb = ps.host.at(p + ".conv_shortcut.bias");
            for (int i = 0; i < co; ++i) c2->b[i] += sb[i];
        } else {
            c2->R = co;
            c2->sc_identity = true;
        }
        if (temb) {
            temb_off[p] = temb_ld;
            temb_ld += co;
        }
        return 0;
    }
    int add_attn(ParamStore& ps, const std::string& p, int c, int head_dim) {
        if (layers.add_norm(ps, p + ".group_norm", c)) return 1;
        if (layers.add_qkv(ps, p, c, head_dim)) return 1;
        if (layers.add_conv(ps, p + ".to_out.0", c, c, 1)) return 1;
        ConvLayer* o = layers.get_conv(p + ".to_out.0");
        o->R = c;
        o->sc_identity = true;
        return 0;
    }
};

static int pad16(int c) { return (c + 15) / 16 * 16; }

}  // namespace rldm

using namespace rldm;

// =================================================================================================================
// UNet2DModel
// =================================================================================================================
struct rldm_unet {
    rldm_unet_config cfg;
    ParamStore params;
    NetCommon net;
    int temb_dim = 0;                               // time_embed_dim
    DevBuf w1, b1, w2, b2, wp, bp;                  // fp32 time-embedding weights
    std::map<int, std::unique_ptr<Plan>> plans;     // per batch size
    DevBuf t_dev, temb_tab;                         // scratch for rldm_unet_forward
    DevBuf temb_scratch;                            // hidden layers of the time-embedding MLP (launch_temb)
    uint64_t generation = 0;                        // bumped whenever the device weights are (re)built: samplers re-plan
    int temb_rows_cap = 0;
    int plan_flags = 0;                             // rldm_unet_set_plan_flags: routing options of the rldm_unet_forward plans
    std::map<int, bool> trunk_checked;              // batch -> the plan's persistent launches passed their self-check once
    std::vector<std::string> resnet_order;

    int levels() const { return cfg.num_levels; }
};

static void unet_expect(rldm_unet* m) {
    const auto& c = m->cfg;
    ParamStore& ps = m->params;
    const int* boc = c.block_out_channels;
    const int L = c.num_levels;
    const int temb = boc[0] * 4;
    ps.expect_conv("conv_in", boc[0], c.in_channels, 3);
    ps.expect_lin("time_embedding.linear_1", temb, boc[0]);
    ps.expect_lin("time_embedding.linear_2", temb, temb);
    int out = boc[0];
    for (int i = 0; i < L; ++i) {
        int cin = out;
        out = boc[i];
        for (int j = 0; j < c.layers_per_block; ++j) {
            ps.expect_resnet("down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? cin : out, out, temb);
            if (c.down_attn[i]) ps.expect_attn("down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), out);
        }
        if (i != L - 1) ps.expect_conv("down_blocks." + std::to_string(i) + ".downsamplers.0.conv", out, out, 3);
    }
    const int cm = boc[L - 1];
    ps.expect_resnet("mid_block.resnets.0", cm, cm, temb);
    if (c.mid_attention) ps.expect_attn("mid_block.attentions.0", cm);
    ps.expect_resnet("mid_block.resnets.1", cm, cm, temb);
    out = boc[L - 1];
    for (int i = 0; i < L; ++i) {
        const int prev = out;
        out = boc[L - 1 - i];
        const int inp = boc[L - 1 - std::min(i + 1, L - 1)];
        const int n = c.layers_per_block + 1;
        for (int j = 0; j < n; ++j) {
            const int skip = (j == n - 1) ? inp : out;
            const int rin = (j == 0) ? prev : out;
            ps.expect_resnet("up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), rin + skip, out, temb);
            if (c.up_attn[i]) ps.expect_attn("up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), out);
        }
        if (i != L - 1) ps.expect_conv("up_blocks." + std::to_string(i) + ".upsamplers.0.conv", out, out, 3);
    }
    ps.expect_norm("conv_norm_out", boc[0]);
    ps.expect_conv("conv_out", c.out_channels, boc[0], 3);
}

static int unet_build_layers(rldm_unet* m) {
    const auto& c = m->cfg;
    ParamStore& ps = m->params;
    NetCommon& net = m->net;
    net = NetCommon();
    net.groups = c.norm_num_groups;
    net.eps = c.norm_eps;
    const int* boc = c.block_out_channels;
    const int L = c.num_levels;
    const int hd = c.attention_head_dim;
    m->temb_dim = boc[0] * 4;
    if (net.layers.add_conv(ps, "conv_in", boc[0], c.in_channels, 3)) return 1;
    int out = boc[0];
    for (int i = 0; i < L; ++i) {
        int cin = out;
        out = boc[i];
        for (int j = 0; j < c.layers_per_block; ++j) {
            const std::string r = "down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j);
            if (net.add_resnet(ps, r, j == 0 ? cin : out, out, true)) return 1;
            m->resnet_order.push_back(r);
            if (c.down_attn[i] && net.add_attn(ps, "down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), out, hd)) return 1;
        }
        if (i != L - 1 && net.layers.add_conv(ps, "down_blocks." + std::to_string(i) + ".downsamplers.0.conv", out, out, 3)) return 1;
    }
    const int cm = boc[L - 1];
    if (net.add_resnet(ps, "mid_block.resnets.0", cm, cm, true)) return 1;
    m->resnet_order.push_back("mid_block.resnets.0");
    if (c.mid_attention && net.add_attn(ps, "mid_block.attentions.0", cm, hd)) return 1;
    if (net.add_resnet(ps, "mid_block.resnets.1", cm, cm, true)) return 1;
    m->resnet_order.push_back("mid_block.resnets.1");
    out = boc[L - 1];
    for (int i = 0; i < L; ++i) {
        const int prev = out;
        out = boc[L - 1 - i];
        const int inp = boc[L - 1 - std::min(i + 1, L - 1)];
        const int n = c.layers_per_block + 1;
        for (int j = 0; j < n; ++j) {
            const int skip = (j == n - 1) ? inp : out;
            const int rin = (j == 0) ? prev : out;
            const std::string r = "up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j);
            if (net.add_resnet(ps, r, rin + skip, out, true)) return 1;
            m->resnet_order.push_back(r);
            if (c.up_attn[i] && net.add_attn(ps, "up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), out, hd)) return 1;
        }
        if (i != L - 1 && net.layers.add_conv(ps, "up_blocks." + std::to_string(i) + ".upsamplers.0.conv", out, out, 3)) return 1;
    }
    if (net.layers.add_norm(ps, "conv_norm_out", boc[0])) return 1;
    if (net.layers.add_conv(ps, "conv_out", c.out_channels, boc[0], 3)) return 1;

    // time-embedding weights (fp32): linear_1, linear_2, all time_emb_proj concatenated in resnet order
    const int D = m->temb_dim;
    if (upload(m->w1, ps.host.at("time_embedding.linear_1.weight").data(), (size_t)D * boc[0] * 4)) return 1;
    if (upload(m->b1, ps.host.at("time_embedding.linear_1.bias").data(), (size_t)D * 4)) return 1;
    if (upload(m->w2, ps.host.at("time_embedding.linear_2.weight").data(), (size_t)D * D * 4)) return 1;
    if (upload(m->b2, ps.host.at("time_embedding.linear_2.bias").data(), (size_t)D * 4)) return 1;
    std::vector<float> wp((size_t)net.temb_ld * D), bp(net.temb_ld);
    for (auto& r : m->resnet_order) {
        const int off = net.temb_off.at(r);
        const auto& w = ps.host.at(r + ".time_emb_proj.weight");
        const auto& b = ps.host.at(r + ".time_emb_proj.bias");
        std::copy(w.begin(), w.end(), wp.begin() + (size_t)off * D);
        std::copy(b.begin(), b.end(), bp.begin() + off);
    }
    if (upload(m->wp, wp.data(), wp.size() * 4)) return 1;
    if (upload(m->bp, bp.data(), bp.size() * 4)) return 1;
    return 0;
}

// one pass of the UNet walk (dry: sizes only)
static int unet_walk(rldm_unet* m, Builder& b, int B) {
    const auto& c = m->cfg;
    NetCommon& net = m->net;
    const int L = c.num_levels;
    const int W = c.sample_w, H = c.sample_h;
    const int Cpad = pad16(c.in_channels);
    Plan* plan = b.plan;

    Tensor xin = b.make(B, W, H, Cpad);
    b.note_launch();
    if (!b.dry) {
        bf16_t* dst = b.tptr(xin);
        plan->io.xin = dst;
        plan->io.xin_ld = Cpad;
        plan->ops.push_back({[plan, dst, B, W, H, Cpad](hipStream_t s) {
            if (plan->io.pack_fused) return 0;          // (the sampler packed x_T; afterwards conv_out's epilogue writes this tensor)
            PackInputParams p{};
            p.x = plan->io.sample; p.cx = plan->io.sample_channels; p.scale = plan->io.sample_scale;
            p.pos_encoding = plan->io.pos_encoding;
            p.cond = plan->io.cond; p.cc = plan->io.cond_channels;
            p.B = B; p.W = W; p.H = H; p.Cpad = Cpad;
            p.out = dst;
            p.step_inc = plan->io.step_inc;
            return launch_pack_input(p, s);
        }, "pack_input_kernel", 0.0, (double)B * W * H * (Cpad * 2.0 + 4.0 * 5), [plan]() { return !plan->io.pack_fused; }});
    }
    Tensor h;
    {
        ConvArgs a;
        a.layer = net.layers.get_conv("conv_in");
        a.x0 = xin;
        a.want_stats = true;
        a.first_of_step = true;
        if (b.conv(a, &h)) return 1;
        // (xin is never released: with the pack launch fused away it must survive from conv_out's epilogue to the next step's conv_in)
    }
    std::vector<Tensor> skips;
    b.retain(h);
    skips.push_back(h);
    for (int i = 0; i < L; ++i) {
        for (int j = 0; j < c.layers_per_block; ++j) {
            Tensor o;
            if (net.resnet(b, "down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), h, Tensor(), &o)) return 1;
            h = o;
            if (c.down_attn[i]) {
                if (net.attention(b, "down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), h, &o)) return 1;
                h = o;
            }
            b.retain(h);
            skips.push_back(h);
        }
        if (i != L - 1) {
            ConvArgs a;
            a.layer = net.layers.get_conv("down_blocks." + std::to_string(i) + ".downsamplers.0.conv");
            a.x0 = h;
            a.stride = 2;
            a.want_stats = true;
            Tensor o;
            if (b.conv(a, &o)) return 1;
            b.release(h);
            h = o;
            b.retain(h);
            skips.push_back(h);
        }
    }
    {
        Tensor o;
        if (net.resnet(b, "mid_block.resnets.0", h, Tensor(), &o)) return 1;
        h = o;
        if (c.mid_attention) {
            if (net.attention(b, "mid_block.attentions.0", h, &o)) return 1;
            h = o;
        }
        if (net.resnet(b, "mid_block.resnets.1", h, Tensor(), &o)) return 1;
        h = o;
    }
    for (int i = 0; i < L; ++i) {
        for (int j = 0; j < c.layers_per_block + 1; ++j) {
            RLDM_REQUIRE(!skips.empty(), "internal: skip stack underflow");
            Tensor sk = skips.back();
            skips.pop_back();
            Tensor o;
            if (net.resnet(b, "up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), h, sk, &o)) return 1;
            h = o;
            if (c.up_attn[i]) {
                if (net.attention(b, "up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), h, &o)) return 1;
                h = o;
            }
        }
        if (i != L - 1) {
            ConvArgs a;
            a.layer = net.layers.get_conv("up_blocks." + std::to_string(i) + ".upsamplers.0.conv");
            a.x0 = h;
            a.up = 2;
            a.want_stats = true;
            Tensor o;
            if (b.conv(a, &o)) return 1;
            b.release(h);
            h = o;
        }
    }
    RLDM_REQUIRE(skips.empty(), "internal: skip stack not drained");
    {
        ConvArgs a;
        a.layer = net.layers.get_conv("conv_out");
        a.x0 = h;
        a.gn = net.layers.get_norm("conv_norm_out");
        a.eps = net.eps; a.groups = net.groups; a.silu = 1;
        a.out_f32_nchw = true;
        Tensor none;
        if (b.conv(a, &none)) return 1;
        b.release(h);
    }
    return 0;
}

static int unet_make_plan(rldm_unet* m, int B, std::unique_ptr<Plan>* out, int* launches = nullptr, int flags = 0) {
    auto plan = std::make_unique<Plan>();
    plan->B = B; plan->W = m->cfg.sample_w; plan->H = m->cfg.sample_h;
    plan->flags = flags;
    if (build_plan(plan.get(), m->net.temb_ld, [&](Builder& b) { return unet_walk(m, b, B); }, launches)) return 1;
    *out = std::move(plan);
    return 0;
}

// time-embedding table for `rows` timesteps (host floats) into `tab`
static int unet_temb(rldm_unet* m, const float* t_dev, int rows, float* tab, hipStream_t s) {
    TembParams p;
    p.t = t_dev; p.rows = rows;
    p.dim0 = m->cfg.block_out_channels[0]; p.D = m->temb_dim; p.total = m->net.temb_ld;
    p.w1 = m->w1.as<float>(); p.b1 = m->b1.as<float>();
    p.w2 = m->w2.as<float>(); p.b2 = m->b2.as<float>();
    p.wp = m->wp.as<float>(); p.bp = m->bp.as<float>();
    p.out = tab;
    const size_t need = (size_t)2 * rows * m->temb_dim * sizeof(float);
    if (m->temb_scratch.bytes < need && m->temb_scratch.alloc(need)) return 1;
    p.scratch = m->temb_scratch.as<float>();
    p.flip_sin_to_cos = m->cfg.flip_sin_to_cos; p.freq_shift = m->cfg.freq_shift;
    return launch_temb(p, s);
}

// =================================================================================================================
// VAE
// =================================================================================================================
struct rldm_vae {
    uint64_t generation = 0;                        // see rldm_unet::generation
    rldm_vae_config cfg;
    ParamStore params;
    NetCommon net;
    struct Key { int B, w, h, enc; bool operator<(const Key& o) const { return std::tie(B, w, h, enc) < std::tie(o.B, o.w, o.h, o.enc); } };
    std::map<Key, std::unique_ptr<Plan>> plans;
};

static void vae_expect(rldm_vae* m) {
    const auto& c = m->cfg;
    ParamStore& ps = m->params;
    const int L = c.num_levels;
    std::vector<int> chs(L);
    for (int i = 0; i < L; ++i) chs[i] = c.ch * c.ch_mult[i];
    ps.expect_conv("encoder.conv_in", c.ch, c.in_channels, 3);
    int cin = c.ch;
    for (int i = 0; i < L; ++i) {
        for (int j = 0; j < c.num_res_blocks; ++j) {
            ps.expect_resnet("encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), cin, chs[i], 0);
            cin = chs[i];
        }
        if (i != L - 1) ps.expect_conv("encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv", cin, cin, 3);
    }
    ps.expect_resnet("encoder.mid_block.resnets.0", cin, cin, 0);
    ps.expect_resnet("encoder.mid_block.resnets.1", cin, cin, 0);
    ps.expect_norm("encoder.conv_norm_out", cin);
    ps.expect_conv("encoder.conv_out", c.double_z ? 2 * c.z_channels : c.z_channels, cin, 3);
    cin = chs[L - 1];
    ps.expect_conv("decoder.conv_in", cin, c.z_channels, 3);
    ps.expect_resnet("decoder.mid_block.resnets.0", cin, cin, 0);
    ps.expect_resnet("decoder.mid_block.resnets.1", cin, cin, 0);
    for (int i = 0; i < L; ++i) {
        const int cout = chs[L - 1 - i];
        for (int j = 0; j < c.num_res_blocks + 1; ++j) {
            ps.expect_resnet("decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), cin, cout, 0);
            cin = cout;
        }
        if (i != L - 1) ps.expect_conv("decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", cin, cin, 3);
    }
    ps.expect_norm("decoder.conv_norm_out", cin);
    ps.expect_conv("decoder.conv_out", c.out_channels, cin, 3);
}

static int vae_build_layers(rldm_vae* m) {
    const auto& c = m->cfg;
    ParamStore& ps = m->params;
    NetCommon& net = m->net;
    net = NetCommon();
    net.groups = c.norm_num_groups;
    net.eps = c.norm_eps;
    const int L = c.num_levels;
    std::vector<int> chs(L);
    for (int i = 0; i < L; ++i) chs[i] = c.ch * c.ch_mult[i];
    if (net.layers.add_conv(ps, "encoder.conv_in", c.ch, c.in_channels, 3)) return 1;
    int cin = c.ch;
    for (int i = 0; i < L; ++i) {
        for (int j = 0; j < c.num_res_blocks; ++j) {
            if (net.add_resnet(ps, "encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), cin, chs[i], false)) return 1;
            cin = chs[i];
        }
        if (i != L - 1 && net.layers.add_conv(ps, "encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv", cin, cin, 3)) return 1;
    }
    if (net.add_resnet(ps, "encoder.mid_block.resnets.0", cin, cin, false)) return 1;
    if (net.add_resnet(ps, "encoder.mid_block.resnets.1", cin, cin, false)) return 1;
    if (net.layers.add_norm(ps, "encoder.conv_norm_out", cin)) return 1;
    if (net.layers.add_conv(ps, "encoder.conv_out", c.double_z ? 2 * c.z_channels : c.z_channels, cin, 3)) return 1;
    cin = chs[L - 1];
    if (net.layers.add_conv(ps, "decoder.conv_in", cin, c.z_channels, 3)) return 1;
    if (net.add_resnet(ps, "decoder.mid_block.resnets.0", cin, cin, false)) return 1;
    if (net.add_resnet(ps, "decoder.mid_block.resnets.1", cin, cin, false)) return 1;
    for (int i = 0; i < L; ++i) {
        const int cout = chs[L - 1 - i];
        for (int j = 0; j < c.num_res_blocks + 1; ++j) {
            if (net.add_resnet(ps, "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), cin, cout, false)) return 1;
            cin = cout;
        }
        if (i != L - 1 && net.layers.add_conv(ps, "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", cin, cin, 3)) return 1;
    }
    if (net.layers.add_norm(ps, "decoder.conv_norm_out", cin)) return 1;
    if (net.layers.add_conv(ps, "decoder.conv_out", c.out_channels, cin, 3)) return 1;
    return 0;
}

static void push_pack_input(Builder& b, Tensor xin) {
    b.note_launch();
    if (b.dry) return;
    Plan* plan = b.plan;
    bf16_t* dst = b.tptr(xin);
    const int B = xin.B, W = xin.W, H = xin.H, Cpad = xin.C;
    plan->ops.push_back({[plan, dst, B, W, H, Cpad](hipStream_t s) {
        PackInputParams p{};
        p.x = plan->io.sample; p.cx = plan->io.sample_channels; p.scale = plan->io.sample_scale;
        p.pos_encoding = 0; p.cond = nullptr; p.cc = 0;
        p.B = B; p.W = W; p.H = H; p.Cpad = Cpad;
        p.out = dst;
        return launch_pack_input(p, s);
    }, "pack_input_kernel", 0.0, (double)B * W * H * (Cpad * 2.0 + 4.0 * 4)});
}

static int vae_walk_decode(rldm_vae* m, Builder& b, int B, int w, int h) {
    const auto& c = m->cfg;
    NetCommon& net = m->net;
    const int L = c.num_levels;
    Tensor xin = b.make(B, w, h, pad16(c.z_channels));
    push_pack_input(b, xin);
    Tensor t;
    ConvArgs a;
    a.layer = net.layers.get_conv("decoder.conv_in");
    a.x0 = xin;
    a.want_stats = true;
    if (b.conv(a, &t)) return 1;
    b.release(xin);
    Tensor o;
    if (net.resnet(b, "decoder.mid_block.resnets.0", t, Tensor(), &o)) return 1;
    t = o;
    if (net.resnet(b, "decoder.mid_block.resnets.1", t, Tensor(), &o)) return 1;
    t = o;
    for (int i = 0; i < L; ++i) {
        for (int j = 0; j < c.num_res_blocks + 1; ++j) {
            if (net.resnet(b, "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), t, Tensor(), &o)) return 1;
            t = o;
        }
        if (i != L - 1) {
            ConvArgs u;
            u.layer = net.layers.get_conv("decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv");
            u.x0 = t;
            u.up = 2;
            u.want_stats = true;
            if (b.conv(u, &o)) return 1;
            b.release(t);
            t = o;
        }
    }
    ConvArgs co;
    co.layer = net.layers.get_conv("decoder.conv_out");
    co.x0 = t;
    co.gn = net.layers.get_norm("decoder.conv_norm_out");
    co.eps = net.eps; co.groups = net.groups; co.silu = 1;
    co.out_f32_nchw = true;
    Tensor none;
    if (b.conv(co, &none)) return 1;
    b.release(t);
    return 0;
}

static int vae_walk_encode(rldm_vae* m, Builder& b, int B, int w, int h) {
    const auto& c = m->cfg;
    NetCommon& net = m->net;
    const int L = c.num_levels;
    Tensor xin = b.make(B, w, h, pad16(c.in_channels));
    push_pack_input(b, xin);
    Tensor t;
    ConvArgs a;
    a.layer = net.layers.get_conv("encoder.conv_in");
    a.x0 = xin;
    a.want_stats = true;
    if (b.conv(a, &t)) return 1;
    b.release(xin);
    Tensor o;
    for (int i = 0; i < L; ++i) {
        for (int j = 0; j < c.num_res_blocks; ++j) {
            if (net.resnet(b, "encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), t, Tensor(), &o)) return 1;
            t = o;
        }
        if (i != L - 1) {
            ConvArgs d;
            d.layer = net.layers.get_conv("encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv");
            d.x0 = t;
            d.stride = 2;
            d.want_stats = true;
            d.pad_mode = 1;                          // end-only pad: ldm/utils.py:109-111, model.py:164-172
            if (b.conv(d, &o)) return 1;
            b.release(t);
            t = o;
        }
    }
    if (net.resnet(b, "encoder.mid_block.resnets.0", t, Tensor(), &o)) return 1;
    t = o;
    if (net.resnet(b, "encoder.mid_block.resnets.1", t, Tensor(), &o)) return 1;
    t = o;
    ConvArgs co;
    co.layer = net.layers.get_conv("encoder.conv_out");
    co.x0 = t;
    co.gn = net.layers.get_norm("encoder.conv_norm_out");
    co.eps = net.eps; co.groups = net.groups; co.silu = 1;
    co.out_f32_nchw = true;
    Tensor none;
    if (b.conv(co, &none)) return 1;
    b.release(t);
    return 0;
}

static int vae_make_plan(rldm_vae* m, int B, int w, int h, bool enc, std::unique_ptr<Plan>* out, int flags = 0) {
    auto plan = std::make_unique<Plan>();
    plan->B = B; plan->W = w; plan->H = h;
    plan->flags = flags;
    if (build_plan(plan.get(), 0, [&](Builder& b) { return enc ? vae_walk_encode(m, b, B, w, h) : vae_walk_decode(m, b, B, w, h); }))
        return 1;
    *out = std::move(plan);
    return 0;
}

static int vae_get_plan(rldm_vae* m, int B, int w, int h, bool enc, Plan** out) {
    rldm_vae::Key k{B, w, h, enc ? 1 : 0};
    auto it = m->plans.find(k);
    if (it == m->plans.end()) {
        std::unique_ptr<Plan> p;
        if (vae_make_plan(m, B, w, h, enc, &p)) return 1;
        it = m->plans.emplace(k, std::move(p)).first;
    }
    *out = it->second.get();
    return 0;
}

// =================================================================================================================
// sampler
// =================================================================================================================
// One lane = an independent slice of the batch with its own plans, buffers, stream and captured graphs.  Whole samples
// never interact, so the lanes' 50-step chains run concurrently on separate streams: the low-resolution UNet levels
// launch far fewer workgroups than the chip has CUs, and a second (third, fourth) chain fills the idle ones.
struct SamplerLane {
    int b0 = 0, nb = 0;                             // samples [b0, b0 + nb) of the batch
    std::unique_ptr<Plan> uplan;                    // private UNet plan (graph-baked pointers)
    std::unique_ptr<Plan> dplan;                    // private VAE decode plan
    DevBuf x, eps, cond, step, image;
    hipStream_t stream = nullptr;
    hipEvent_t ev_out = nullptr;
    hipGraphExec_t step_graph = nullptr, decode_graph = nullptr;
    int graph_steps = 1;                            // sampler steps captured in step_graph
    bool fused_tail = true;                         // scheduler step in conv_out's epilogue, step index advanced by pack_input
    const float* captured_noise = nullptr;
    long long n_latent = 0, n_image = 0, n_cond = 0;
    int* check_host = nullptr;                      // pinned: the persistent launches' self-check word of the last call (copied behind its
    bool check_pending = false;                     // last launch; read by rldm_sampler_status, the next call or the destructor)
    std::atomic<bool> call_recorded{false};         // ev_out has been recorded at least once (read by other samplers' threads: in_flight)
    ~SamplerLane() {
        if (check_host) (void)hipHostFree(check_host);
        if (step_graph) (void)hipGraphExecDestroy(step_graph);
        if (decode_graph) (void)hipGraphExecDestroy(decode_graph);
        if (ev_out) (void)hipEventDestroy(ev_out);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

struct rldm_sampler {
    rldm_unet* unet = nullptr;
    rldm_vae* vae = nullptr;
    rldm_sampler_config cfg;
    DevBuf coef, temb_tab, t_dev;                   // shared, read-only after creation
    std::vector<std::unique_ptr<SamplerLane>> lanes;
    hipEvent_t ev_in = nullptr;
    long long n_latent = 0, n_image = 0;            // whole batch
    uint64_t unet_gen = 0, vae_gen = 0;             // generations of the weights the lanes' plans and graphs were built on
    int plan_flags = 0;                             // routing options of this sampler's plans (rldm_sampler_config::plan_flags | fall-backs)
    int device = 0;
    std::atomic<hipStream_t> last_caller{nullptr};  // stream of the last rldm_sample call (ordering against other samplers' calls; read
                                                    // by other samplers' host threads under g_samplers_mu -- atomics, not plain fields)
    std::atomic<bool> shared_mark{false};           // another sampler found THIS one in flight beside its own call: drop the persistent
                                                    // launches at the next call instead of meeting that sampler's launches again
    int latched_error = 0;                          // self-check code of a call whose pending word was read while the plans were rebuilt
    int inject_error = 0;                           // tests: rldm_debug_inject_trunk_error (one shot)
    bool has_persistent() const {
        for (auto& ln : lanes)
            if (ln->uplan && ln->uplan->trunk_error.p) return true;
        return false;
    }
    bool in_flight() const {                        // some lane's last call has not finished
        for (auto& ln : lanes)
            if (ln->ev_out && ln->call_recorded && hipEventQuery(ln->ev_out) == hipErrorNotReady) return true;
        return false;
    }
    ~rldm_sampler();
};

// Live samplers of the process: a sampler whose plans hold persistent launches (trunk.hip: 256 co-resident workgroups that wait for
// each other) must have the device to itself while they run.  rldm_sample looks for another sampler's call still in flight on a
// DIFFERENT caller stream (same stream: ordered behind it) and, if there is one, runs -- from then on -- one launch per layer.
static std::mutex g_samplers_mu;
static std::vector<rldm_sampler*> g_samplers;

rldm_sampler::~rldm_sampler() {
    {
        std::lock_guard<std::mutex> lk(g_samplers_mu);
        g_samplers.erase(std::remove(g_samplers.begin(), g_samplers.end(), this), g_samplers.end());
    }
    for (auto& ln : lanes) {                        // the last call of a run is never followed by another: say so here
        if (ln->check_pending && ln->ev_out && hipEventSynchronize(ln->ev_out) == hipSuccess && ln->check_host && *ln->check_host != 0)
            fprintf(stderr, "librangeldm_hip: the LAST rldm_sample call of a sampler failed the self-check of its persistent launches "
                            "(code %d): its images were NaN-marked\n", *ln->check_host);
    }
    lanes.clear();
    if (ev_in) (void)hipEventDestroy(ev_in);
}

static DevBuf g_trace;                 // rldm_debug_graph_trace: 4096 timestamps
static Plan* g_trace_plan = nullptr;

// (rldm_debug_set_flags(1 << 23) at sampler creation keeps the scheduler step and the step counter as launches of their own)

// x_T (just copied into the lane's x) as conv_in's input: once per call when the steps' pack_input launch is fused into conv_out
static int sampler_pack_x(rldm_sampler* s, SamplerLane* ln, hipStream_t st) {
    const PlanIO& io = ln->uplan->io;
    if (!io.pack_fused) return 0;
    const auto& uc = s->unet->cfg;
    PackInputParams p{};
    p.x = io.sample; p.cx = io.sample_channels; p.scale = io.sample_scale;
    p.pos_encoding = io.pos_encoding;
    p.cond = io.cond; p.cc = io.cond_channels;
    p.B = ln->nb; p.W = uc.sample_w; p.H = uc.sample_h; p.Cpad = io.xin_ld;
    p.out = io.xin;
    p.step_inc = nullptr;
    return launch_pack_input(p, st);
}

static int sampler_enqueue_step(rldm_sampler* s, SamplerLane* ln, const float* noise, hipStream_t st) {
    const bool fused = ln->fused_tail;
    if (fused) ln->uplan->io.sch.noise = noise;        // (the rest of io.sch / io.step_inc: sampler_build_plans)
    if ((g_dbg_flags | s->plan_flags) & 8192) {
        if (!g_trace.p) {
            if (g_trace.alloc(4096 * 8)) return 1;
            RLDM_HIP_CHECK(hipMemset(g_trace.p, 0, 4096 * 8));
        }
        g_trace_plan = ln->uplan.get();
        if (ln->uplan->run_stamped(st, g_trace.as<unsigned long long>(), 4096)) return 1;
    } else if (ln->uplan->run(st)) return 1;
    if (fused) return 0;
    SchedParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.mode = (s->cfg.mode == RLDM_SAMPLER_DDIM ? 0 : 1) | (s->cfg.prediction_type << 1);
    sp.coef_table = s->coef.as<float>();
    sp.step_ptr = ln->step.as<int>();
    sp.eps = ln->eps.as<float>();
    sp.x = ln->x.as<float>();
    sp.noise = noise;                               // already offset to this lane's first sample
    sp.noise_step_stride = s->n_latent;             // the caller's tensor is [steps][whole batch][...]
    sp.x_prev = ln->x.as<float>();
    sp.n = ln->n_latent;
    if (launch_sched_step(sp, st)) return 1;
    return launch_step_counter(ln->step.as<int>(), 0, 1, st);
}

// (Re)build everything of a sampler that is baked on the models' device weights: the time-embedding table, every lane's
// UNet / decode plan (weight pointers, packed images) and -- by dropping them -- the captured graphs.  Called at creation
// and again by rldm_sample when rldm_unet_finalize / rldm_vae_finalize ran since (load_state_dict on a model a pipeline
// already sampled with: the old graphs would replay over freed weight buffers).
static int sampler_build_plans(rldm_sampler* s) {
    rldm_unet* unet = s->unet;
    rldm_vae* vae = s->vae;
    RLDM_REQUIRE(unet->params.finalized, "unet not finalized (set_param without rldm_unet_finalize)");
    RLDM_REQUIRE(!vae || vae->params.finalized, "vae not finalized (set_param without rldm_vae_finalize)");
    const auto& uc = unet->cfg;
    const int W = uc.sample_w, H = uc.sample_h;
    for (auto& lnp : s->lanes) RLDM_HIP_CHECK(hipStreamSynchronize(lnp->stream));
    if (s->temb_tab.alloc((size_t)s->cfg.num_steps * unet->net.temb_ld * 4)) return 1;
    if (unet_temb(unet, s->t_dev.as<float>(), s->cfg.num_steps, s->temb_tab.as<float>(), nullptr)) return 1;
    RLDM_HIP_CHECK(hipStreamSynchronize(nullptr));
    for (auto& lnp : s->lanes) {
        SamplerLane* ln = lnp.get();
        if (ln->step_graph) { (void)hipGraphExecDestroy(ln->step_graph); ln->step_graph = nullptr; }
        if (ln->decode_graph) { (void)hipGraphExecDestroy(ln->decode_graph); ln->decode_graph = nullptr; }
        ln->captured_noise = nullptr;
        ln->uplan.reset();
        ln->dplan.reset();
        g_concurrent_plans = (int)s->lanes.size();
        const int prc = unet_make_plan(unet, ln->nb, &ln->uplan, nullptr, s->plan_flags);
        g_concurrent_plans = 1;
        // (the lane streams were synchronised above: a pending self-check word of the previous, asynchronous call has landed -- keep it)
        if (ln->check_pending && ln->check_host && *ln->check_host != 0) s->latched_error = *ln->check_host;
        ln->check_pending = false;
        if (prc) return 1;
        PlanIO& io = ln->uplan->io;
        io.sample = ln->x.as<float>();
        io.sample_channels = uc.out_channels;
        io.pos_encoding = s->cfg.pos_encoding;
        io.cond = s->cfg.cond_channels ? ln->cond.as<float>() : nullptr;
        io.cond_channels = s->cfg.cond_channels;
        io.out = ln->eps.as<float>();
        io.temb = s->temb_tab.as<float>();
        io.step_ptr = ln->step.as<int>();
        io.temb_rows_per_step = 1;
        io.temb_per_sample = 0;
        ln->fused_tail = !((g_dbg_flags | s->plan_flags) & (1 << 23));
        if (ln->fused_tail) {
            // the scheduler step rides in conv_out's epilogue, the step index is advanced by pack_input: 2 launches per step fewer
            SchedFuse& f = io.sch;
            f.coef_table = s->coef.as<float>();
            f.step_ptr = ln->step.as<int>();
            f.x = ln->x.as<float>();
            f.noise = nullptr;                          // per call: sampler_enqueue_step
            f.noise_step_stride = s->n_latent;
            f.x_prev = ln->x.as<float>();
            f.mode = (s->cfg.mode == RLDM_SAMPLER_DDIM ? 0 : 1) | (s->cfg.prediction_type << 1);
            io.step_inc = ln->step.as<int>();
            // ... and the next step's conv_in input: no pack_input launch inside the steps (rldm_debug_set_flags(64) keeps it)
            if (!((g_dbg_flags | s->plan_flags) & 64) && io.xin && io.sample_scale == 1.f) {
                io.pack_fused = true;
                f.pack = io.xin;
                f.pack_ld = io.xin_ld;
            }
        }
        if (vae) {
            if (vae_make_plan(vae, ln->nb, W, H, false, &ln->dplan, s->plan_flags)) return 1;
            PlanIO& d = ln->dplan->io;
            d.sample = ln->x.as<float>();
            d.sample_channels = vae->cfg.z_channels;
            d.sample_scale = 1.0f / vae->cfg.scaling_factor;      // latents / scaling_factor, ldm/pipelines.py:365
            d.out = ln->image.as<float>();
        }
    }
    s->unet_gen = unet->generation;
    s->vae_gen = vae ? vae->generation : 0;
    return 0;
}

static int capture(hipStream_t st, const std::function<int()>& body, hipGraphExec_t* exec) {
    RLDM_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = body();
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(st, &g);
    if (rc) {
        if (g) (void)hipGraphDestroy(g);
        return 1;
    }
    RLDM_HIP_CHECK(e);
    if (*exec) {
        (void)hipGraphExecDestroy(*exec);
        *exec = nullptr;
    }
    RLDM_HIP_CHECK(hipGraphInstantiate(exec, g, nullptr, nullptr, 0));
    RLDM_HIP_CHECK(hipGraphDestroy(g));
    return 0;
}

// =================================================================================================================
// C ABI
// =================================================================================================================
extern "C" {

const char* rldm_last_error(void) { return g_error.c_str(); }

int rldm_device_info(char* name, size_t name_len, int* compute_units) {
    int n = 0;
    RLDM_HIP_CHECK(hipGetDeviceCount(&n));
    RLDM_REQUIRE(n > 0, "no HIP device visible");
    int dev = 0;
    RLDM_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    RLDM_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (name && name_len) {
        strncpy(name, prop.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    RLDM_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
                 std::string("librangeldm_hip is built for gfx950 only; device is ") + prop.gcnArchName);
    return 0;
}

// ---- UNet -------------------------------------------------------------------------------------------------------
int rldm_unet_create(const rldm_unet_config* cfg, rldm_unet** out) {
    RLDM_REQUIRE(cfg && out, "null argument");
    RLDM_REQUIRE(cfg->num_levels >= 1 && cfg->num_levels <= RLDM_MAX_LEVELS, "num_levels out of range");
    RLDM_REQUIRE(cfg->attention_head_dim == 8, "only attention_head_dim == 8 (the UNet2DModel default the reference uses) is supported");
    RLDM_REQUIRE(cfg->in_channels <= 16, "in_channels > 16 not supported");
    for (int i = 0; i < cfg->num_levels; ++i)
        RLDM_REQUIRE(cfg->block_out_channels[i] % 32 == 0 && cfg->block_out_channels[i] <= 512,
                     "block_out_channels must be multiples of 32, <= 512");
    RLDM_REQUIRE((cfg->sample_w >> (cfg->num_levels - 1)) >= 2 && (cfg->sample_h >> (cfg->num_levels - 1)) >= 1 &&
                     cfg->sample_w % (1 << (cfg->num_levels - 1)) == 0 && cfg->sample_h % (1 << (cfg->num_levels - 1)) == 0,
                 "sample_size not divisible by 2^(levels-1)");
    auto* m = new rldm_unet();
    m->cfg = *cfg;
    unet_expect(m);
    *out = m;
    return 0;
}
void rldm_unet_destroy(rldm_unet* m) { delete m; }

int rldm_unet_set_param(rldm_unet* m, const char* name, const float* data, int64_t numel) {
    RLDM_REQUIRE(m && name && data, "null argument");
    m->plans.clear();
    m->params.finalized = false;                    // a live sampler refuses to run until rldm_unet_finalize rebuilt the weights
    return m->params.set(name, data, numel);
}

int rldm_unet_finalize(rldm_unet* m) {
    RLDM_REQUIRE(m, "null argument");
    if (m->params.check_complete()) return 1;
    m->resnet_order.clear();
    m->plans.clear();
    if (unet_build_layers(m)) return 1;
    m->params.finalized = true;
    ++m->generation;
    return 0;
}

static int unet_get_plan(rldm_unet* m, int B, Plan** out) {
    RLDM_REQUIRE(m->params.finalized, "rldm_unet_finalize has not been called");
    auto it = m->plans.find(B);
    if (it == m->plans.end()) {
        std::unique_ptr<Plan> p;
        if (unet_make_plan(m, B, &p, nullptr, m->plan_flags)) return 1;
        it = m->plans.emplace(B, std::move(p)).first;
        m->trunk_checked.erase(B);
    }
    *out = it->second.get();
    return 0;
}

int rldm_unet_forward(rldm_unet* m, const float* sample, const int64_t* timesteps, int nt, int B, float* out, void* stream) {
    RLDM_REQUIRE(m && sample && timesteps && out, "null argument");
    RLDM_REQUIRE(B >= 1 && (nt == 1 || nt == B), "timesteps must have length 1 or B");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    Plan* plan = nullptr;
    if (unet_get_plan(m, B, &plan)) return 1;
    if (m->temb_rows_cap < nt) {
        if (m->t_dev.alloc((size_t)nt * 4)) return 1;
        if (m->temb_tab.alloc((size_t)nt * m->net.temb_ld * 4)) return 1;
        m->temb_rows_cap = nt;
    }
    std::vector<float> tf(nt);
    for (int i = 0; i < nt; ++i) tf[i] = (float)timesteps[i];
    RLDM_HIP_CHECK(hipMemcpyAsync(m->t_dev.p, tf.data(), (size_t)nt * 4, hipMemcpyHostToDevice, st));
    RLDM_HIP_CHECK(hipStreamSynchronize(st));   // tf is a stack-lifetime staging buffer
    if (unet_temb(m, m->t_dev.as<float>(), nt, m->temb_tab.as<float>(), st)) return 1;
    plan->io = PlanIO();
    plan->io.sample = sample;
    plan->io.sample_channels = m->cfg.in_channels;
    plan->io.out = out;
    plan->io.temb = m->temb_tab.as<float>();
    plan->io.step_ptr = nullptr;
    plan->io.temb_rows_per_step = nt;
    plan->io.temb_per_sample = (nt == B && B > 1) ? 1 : 0;
    if (plan->run(st)) return 1;
    if (plan->trunk_error.p && !m->trunk_checked[B]) {
        // first run of a plan with persistent launches: read their self-check word before anybody uses `out` (the sampler does the same
        // at its warm-up step); if it is set, this model's plans are rebuilt as one launch per layer and the forward runs again
        RLDM_HIP_CHECK(hipStreamSynchronize(st));
        int terr = 0;
        RLDM_HIP_CHECK(hipMemcpy(&terr, plan->trunk_error.p, 4, hipMemcpyDeviceToHost));
        if (terr != 0) {
            RLDM_REQUIRE(!((g_dbg_flags | m->plan_flags) & (1 << 24)), "internal: self-check word set without persistent launches");
            fprintf(stderr, "librangeldm_hip: persistent launches of rldm_unet_forward failed their self-check (code %d); this model runs "
                            "one launch per layer from here on\n", terr);
            m->plan_flags |= (1 << 24);
            m->plans.clear();
            return rldm_unet_forward(m, sample, timesteps, nt, B, out, stream);
        }
        m->trunk_checked[B] = true;
    }
    return 0;
}

double rldm_unet_flops(rldm_unet* m, int B) {
    Plan* plan = nullptr;
    if (!m || unet_get_plan(m, B, &plan)) return -1.0;
    return plan->flops;
}

// self-check word of the persistent trunk launches of the batch-B plan (0 fine / no trunk; 1 a wait gave up; 2 a cluster was spread
// over several XCDs); synchronises the device
int rldm_unet_trunk_status(rldm_unet* m, int B) {
    if (!m) return -1;
    auto it = m->plans.find(B);
    if (it == m->plans.end() || !it->second->trunk_error.p) return 0;
    int terr = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(&terr, it->second->trunk_error.p, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return terr;
}

int rldm_unet_set_plan_flags(rldm_unet* m, int flags) {
    RLDM_REQUIRE(m, "null argument");
    if (flags != m->plan_flags) m->plans.clear();
    m->plan_flags = flags;
    return 0;
}

int rldm_unet_num_launches(rldm_unet* m, int B) {
    if (!m || !m->params.finalized) return -1;
    PlanFlagScope scope(m->plan_flags);
    Plan tmp;
    ViewPlan vplan;                     // same three-pass scheme as build_plan, without the real pass
    {
        Builder rec;
        rec.plan = &tmp;
        rec.dry = true;
        rec.recording = true;
        rec.vp = &vplan;
        rec.temb_ld = m->net.temb_ld;
        if (unet_walk(m, rec, B)) return -1;
        vplan.decide();
    }
    Builder dry;
    dry.plan = &tmp;
    dry.dry = true;
    dry.vp = &vplan;
    dry.temb_ld = m->net.temb_ld;
    if (unet_walk(m, dry, B)) return -1;
    dry.flush_trunk();
    return dry.launches + 1;   // + time-embedding kernel
}

// ---- VAE --------------------------------------------------------------------------------------------------------
int rldm_vae_create(const rldm_vae_config* cfg, rldm_vae** out) {
    RLDM_REQUIRE(cfg && out, "null argument");
    RLDM_REQUIRE(cfg->num_levels >= 1 && cfg->num_levels <= RLDM_MAX_LEVELS, "num_levels out of range");
    RLDM_REQUIRE(cfg->in_channels <= 16 && cfg->z_channels <= 16, "in/z channels > 16 not supported");
    for (int i = 0; i < cfg->num_levels; ++i)
        RLDM_REQUIRE((cfg->ch * cfg->ch_mult[i]) % 32 == 0 && cfg->ch * cfg->ch_mult[i] <= 512, "ch*mult must be a multiple of 32, <= 512");
    auto* m = new rldm_vae();
    m->cfg = *cfg;
    vae_expect(m);
    *out = m;
    return 0;
}
void rldm_vae_destroy(rldm_vae* m) { delete m; }
int rldm_vae_set_param(rldm_vae* m, const char* name, const float* data, int64_t numel) {
    RLDM_REQUIRE(m && name && data, "null argument");
    m->plans.clear();
    m->params.finalized = false;
    return m->params.set(name, data, numel);
}
int rldm_vae_finalize(rldm_vae* m) {
    RLDM_REQUIRE(m, "null argument");
    if (m->params.check_complete()) return 1;
    m->plans.clear();
    if (vae_build_layers(m)) return 1;
    m->params.finalized = true;
    ++m->generation;
    return 0;
}

int rldm_vae_decode(rldm_vae* m, const float* z, int B, int latent_w, int latent_h, float* image, void* stream) {
    RLDM_REQUIRE(m && z && image, "null argument");
    RLDM_REQUIRE(m->params.finalized, "rldm_vae_finalize has not been called");
    Plan* plan = nullptr;
    if (vae_get_plan(m, B, latent_w, latent_h, false, &plan)) return 1;
    plan->io = PlanIO();
    plan->io.sample = z;
    plan->io.sample_channels = m->cfg.z_channels;
    plan->io.out = image;
    return plan->run(reinterpret_cast<hipStream_t>(stream));
}

int rldm_vae_encode(rldm_vae* m, const float* x, int B, int w, int h, float* moments, void* stream) {
    RLDM_REQUIRE(m && x && moments, "null argument");
    RLDM_REQUIRE(m->params.finalized, "rldm_vae_finalize has not been called");
    Plan* plan = nullptr;
    if (vae_get_plan(m, B, w, h, true, &plan)) return 1;
    plan->io = PlanIO();
    plan->io.sample = x;
    plan->io.sample_channels = m->cfg.in_channels;
    plan->io.out = moments;
    return plan->run(reinterpret_cast<hipStream_t>(stream));
}

double rldm_vae_decode_flops(rldm_vae* m, int B, int latent_w, int latent_h) {
    Plan* plan = nullptr;
    if (!m || !m->params.finalized || vae_get_plan(m, B, latent_w, latent_h, false, &plan)) return -1.0;
    return plan->flops;
}

int rldm_diag_gaussian_sample(const float* moments, const float* noise, float scale, int B, int z, int spatial,
                              float* out, void* stream) {
    RLDM_REQUIRE(moments && noise && out, "null argument");
    return launch_diag_gaussian(moments, noise, scale, B, z, spatial, out, reinterpret_cast<hipStream_t>(stream));
}

// ---- scheduler steps ----------------------------------------------------------------------------------------------
static int sched_step(int mode, const float coef[5], const float* eps, const float* x, const float* noise, float* x_prev,
                      int64_t n, void* stream) {
    RLDM_REQUIRE(coef && eps && x && x_prev, "null argument");
    RLDM_REQUIRE(coef[4] == 0.f || noise != nullptr, "sigma != 0 requires a noise tensor");
    SchedParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.mode = mode;
    for (int i = 0; i < 5; ++i) sp.coef[i] = coef[i];
    sp.eps = eps; sp.x = x; sp.noise = noise; sp.x_prev = x_prev; sp.n = n;
    return launch_sched_step(sp, reinterpret_cast<hipStream_t>(stream));
}
int rldm_sched_ddim_step(const float coef[5], const float* eps, const float* x, const float* noise, float* x_prev,
                         int64_t n, void* stream) {
    return sched_step(0, coef, eps, x, noise, x_prev, n, stream);
}
int rldm_sched_ddpm_step(const float coef[5], const float* eps, const float* x, const float* noise, float* x_prev,
                         int64_t n, void* stream) {
    return sched_step(1, coef, eps, x, noise, x_prev, n, stream);
}
int rldm_sched_step(int sampler_mode, int prediction_type, const float coef[5], const float* model_output, const float* x,
                    const float* noise, float* x_prev, int64_t n, void* stream) {
    RLDM_REQUIRE(sampler_mode == RLDM_SAMPLER_DDIM || sampler_mode == RLDM_SAMPLER_DDPM, "unknown sampler mode");
    RLDM_REQUIRE(prediction_type >= RLDM_PRED_EPSILON && prediction_type <= RLDM_PRED_SAMPLE, "unknown prediction type");
    return sched_step((sampler_mode == RLDM_SAMPLER_DDIM ? 0 : 1) | (prediction_type << 1), coef, model_output, x, noise, x_prev, n, stream);
}
int rldm_sched_add_noise(const float* x0, const float* noise, const float* sqrt_alpha, const float* sqrt_beta, int B,
                         int64_t per_sample, float* out, void* stream) {
    RLDM_REQUIRE(x0 && noise && sqrt_alpha && sqrt_beta && out, "null argument");
    return launch_add_noise(x0, noise, sqrt_alpha, sqrt_beta, B, per_sample, out, reinterpret_cast<hipStream_t>(stream));
}

// ---- sampler ------------------------------------------------------------------------------------------------------
static int sampler_num_lanes(int batch, int W, int H) {
    // ONE chain up to batch 31: chains of fewer than 16 samples lose more to the fixed cost of every launch than they gain
    // from running side by side (batch 16: two chains of 8 measured 165 img/s against 194 for a single chain).  From batch
    // 32 on, chains of >= 16 samples overlap usefully (the launches of one chain fill the CUs the other leaves idle in its
    // prologues / low-resolution levels): 2 x 16 = 241 img/s against 229 for one chain of 32, 3 x 16 = 264 against 1 x 48,
    // 2 x 32 = 267 against 264 for one chain of 64; four chains were slower again (238).  RLDM_LANES=n overrides.
    // The unit is WORK, not samples: a chain wants >= 16 x 4096 network-input pixels (16 KITTI latents of 256 x 16).  nuScenes
    // latents are 256 x 8: 32 of them as 2 x 16 measured 301 img/s against 358 for one chain, so they split from 64 on.
    const long long px = (long long)batch * W * H, unit = 16LL * 4096;
    // Pixel-space networks (RangeDM, 1024 x 64 per sample) fill the chip with every launch of a single sample: 2 chains of 2
    // measured 91 img/s against 93.5 for one chain of 4, so only latent-space samplers split.
    int want = (px >= 2 * unit && W * H <= 8192) ? (int)std::min<long long>(3, px / unit) : 1;
    if (const char* e = getenv("RLDM_LANES")) want = atoi(e);
    if (want <= 0) want = 1;
    want = std::max(1, std::min(want, batch));
    while (batch % want) --want;
    return want;
}

int rldm_sampler_create(rldm_unet* unet, rldm_vae* vae, const rldm_sampler_config* cfg, rldm_sampler** out) {
    RLDM_REQUIRE(unet && cfg && out, "null argument");
    RLDM_REQUIRE(unet->params.finalized, "unet not finalized");
    RLDM_REQUIRE(!vae || vae->params.finalized, "vae not finalized");
    RLDM_REQUIRE(cfg->batch >= 1 && cfg->num_steps >= 1 && cfg->coef && cfg->timesteps, "bad sampler config");
    RLDM_REQUIRE(cfg->prediction_type >= RLDM_PRED_EPSILON && cfg->prediction_type <= RLDM_PRED_SAMPLE, "bad sampler config: prediction_type");
    const auto& uc = unet->cfg;
    RLDM_REQUIRE(uc.out_channels + (cfg->pos_encoding ? 1 : 0) + cfg->cond_channels == uc.in_channels,
                 "unet.in_channels != out_channels + pos_encoding + cond_channels (ldm/pipelines.py:351,480)");
    RLDM_REQUIRE(!vae || vae->cfg.z_channels == uc.out_channels, "vae latent channels != unet out_channels");
    auto s = std::make_unique<rldm_sampler>();
    s->unet = unet;
    s->vae = vae;
    s->cfg = *cfg;
    s->cfg.coef = nullptr;
    s->cfg.timesteps = nullptr;
    s->plan_flags = cfg->plan_flags;
    RLDM_HIP_CHECK(hipGetDevice(&s->device));
    const int B = cfg->batch, W = uc.sample_w, H = uc.sample_h;
    const long long per_latent = (long long)uc.out_channels * W * H;
    const long long per_cond = (long long)cfg->cond_channels * W * H;
    long long per_image = 0;
    if (vae) {
        const int f = 1 << (vae->cfg.num_levels - 1);
        per_image = (long long)vae->cfg.out_channels * (W * f) * (H * f);
    }
    s->n_latent = B * per_latent;
    s->n_image = B * per_image;
    RLDM_HIP_CHECK(hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming));
    if (upload(s->coef, cfg->coef, (size_t)cfg->num_steps * 5 * 4)) return 1;
    std::vector<float> tf(cfg->num_steps);
    for (int i = 0; i < cfg->num_steps; ++i) tf[i] = (float)cfg->timesteps[i];
    if (upload(s->t_dev, tf.data(), tf.size() * 4)) return 1;
    const int nl = sampler_num_lanes(B, W, H);
    for (int l = 0; l < nl; ++l) {
        auto ln = std::make_unique<SamplerLane>();
        ln->nb = B / nl;
        ln->b0 = l * ln->nb;
        ln->n_latent = ln->nb * per_latent;
        ln->n_image = ln->nb * per_image;
        ln->n_cond = ln->nb * per_cond;
        RLDM_HIP_CHECK(hipStreamCreateWithFlags(&ln->stream, hipStreamNonBlocking));
        RLDM_HIP_CHECK(hipEventCreateWithFlags(&ln->ev_out, hipEventDisableTiming));
        if (ln->x.alloc(ln->n_latent * 4) || ln->eps.alloc(ln->n_latent * 4) || ln->step.alloc(64)) return 1;
        if (cfg->cond_channels && ln->cond.alloc((size_t)ln->n_cond * 4)) return 1;
        if (vae && ln->image.alloc(ln->n_image * 4)) return 1;
        s->lanes.push_back(std::move(ln));
    }
    if (sampler_build_plans(s.get())) return 1;
    {
        std::lock_guard<std::mutex> lk(g_samplers_mu);
        g_samplers.push_back(s.get());
    }
    *out = s.release();
    return 0;
}
void rldm_sampler_destroy(rldm_sampler* s) { delete s; }

// this sampler's plans again, every layer a launch of its own (same kernels, same tiles: rldm_debug_set_flags(1 << 24), scoped to it)
static int sampler_drop_persistent(rldm_sampler* s, const char* why, int code) {
    fprintf(stderr, "librangeldm_hip: %s (code %d); this sampler runs one launch per layer from here on\n", why, code);
    s->plan_flags |= (1 << 24);
    return sampler_build_plans(s);
}

// Self-check of the LAST rldm_sample call: waits for the call (the lanes' final events), then reads the word its persistent launches
// left.  0: the call's outputs are valid.  Non-zero (1 a cluster wait gave up, 2 a cluster was spread over several XCDs): the outputs
// were NaN-marked, rldm_last_error says why, and the sampler has already rebuilt itself without persistent launches -- call again.
int rldm_sampler_status(rldm_sampler* s) {
    RLDM_REQUIRE(s, "null argument");
    int code = 0;
    for (auto& lnp : s->lanes) {
        SamplerLane* ln = lnp.get();
        if (!ln->check_pending) continue;
        RLDM_HIP_CHECK(hipEventSynchronize(ln->ev_out));
        ln->check_pending = false;
        if (*ln->check_host != 0) code = *ln->check_host;
    }
    if (s->latched_error) { code = s->latched_error; s->latched_error = 0; }
    if (code == 0) return 0;
    if (sampler_drop_persistent(s, "a persistent launch of the last rldm_sample call failed its self-check", code)) return -1;
    set_error("a persistent launch of this rldm_sample call failed its self-check (code " + std::to_string(code) +
              ": 1 a cluster wait gave up -- the GPU was shared with other work --, 2 a cluster was spread over several XCDs); the call's "
              "outputs are invalid and NaN-marked; the sampler now runs one launch per layer: call rldm_sample again");
    return code;
}

int rldm_debug_inject_trunk_error(rldm_sampler* s, int code) {
    RLDM_REQUIRE(s, "null argument");
    s->inject_error = code;
    return 0;
}

int rldm_sample(rldm_sampler* s, const float* x_T, const float* step_noise, const float* cond, float* images,
                float* latents_out, void* stream) {
    RLDM_REQUIRE(s && x_T, "null argument");
    RLDM_REQUIRE(images || latents_out, "no output requested");
    RLDM_REQUIRE((s->cfg.cond_channels == 0) == (cond == nullptr), "cond tensor does not match sampler.cond_channels");
    RLDM_REQUIRE(s->cfg.mode == RLDM_SAMPLER_DDIM || step_noise != nullptr, "DDPM sampling needs step_noise");
    hipStream_t caller = reinterpret_cast<hipStream_t>(stream);
    if (s->unet_gen != s->unet->generation || (s->vae && s->vae_gen != s->vae->generation) || !s->unet->params.finalized ||
        (s->vae && !s->vae->params.finalized)) {
        if (sampler_build_plans(s)) return 1;       // the models were reloaded since the graphs were captured
    }
    // a host that did not ask rldm_sampler_status about the previous call learns here that it failed (its outputs were NaN-marked);
    // the sampler has rebuilt itself without persistent launches by the time this returns, so the retry is valid
    for (auto& lnp : s->lanes) {
        SamplerLane* ln = lnp.get();
        if (ln->check_pending && hipEventQuery(ln->ev_out) == hipSuccess) {
            ln->check_pending = false;
            const int code = *ln->check_host;
            if (code != 0) {
                if (sampler_drop_persistent(s, "a persistent launch of the PREVIOUS rldm_sample call failed its self-check", code)) return 1;
                RLDM_REQUIRE(false, "a persistent launch of the PREVIOUS rldm_sample call failed its self-check (code " + std::to_string(code) +
                                        "): its outputs were invalid (NaN-marked); the sampler now runs one launch per layer: call again");
            }
        }
    }
    // persistent launches need the device to themselves: another sampler's call still in flight on a different stream ends them here
    if (s->has_persistent()) {
        bool shared = s->shared_mark.exchange(false);
        {
            std::lock_guard<std::mutex> lk(g_samplers_mu);
            for (rldm_sampler* o : g_samplers)
                if (o != s && o->device == s->device && o->last_caller.load() != caller && o->in_flight()) {
                    shared = true;
                    o->shared_mark.store(true);     // (the peer keeps its spin-waiting clusters for the call in flight; not for the next)
                }
        }
        if (shared && sampler_drop_persistent(s, "another sampler is running on this device on a different stream: persistent launches "
                                                 "need the chip to themselves", 0)) return 1;
    }
    if (s->latched_error) {                         // found while the plans were rebuilt (sampler_build_plans): the previous call failed
        const int code = s->latched_error;
        s->latched_error = 0;
        if (!(s->plan_flags & (1 << 24)) && sampler_drop_persistent(s, "a persistent launch of the PREVIOUS rldm_sample call failed its self-check", code)) return 1;
        RLDM_REQUIRE(false, "a persistent launch of the PREVIOUS rldm_sample call failed its self-check (code " + std::to_string(code) +
                                "): its outputs were invalid (NaN-marked); the sampler now runs one launch per layer: call again");
    }
    s->last_caller.store(caller);
    RLDM_HIP_CHECK(hipEventRecord(s->ev_in, caller));
    // per lane: inputs, (first call) eager warm step + graph capture
    for (auto& lnp : s->lanes) {
        SamplerLane* ln = lnp.get();
        hipStream_t st = ln->stream;
        const size_t lat_off = (size_t)ln->b0 * (ln->n_latent / ln->nb);
        RLDM_HIP_CHECK(hipStreamWaitEvent(st, s->ev_in, 0));
        RLDM_HIP_CHECK(hipMemcpyAsync(ln->x.p, x_T + lat_off, ln->n_latent * 4, hipMemcpyDeviceToDevice, st));
        if (cond)
            RLDM_HIP_CHECK(hipMemcpyAsync(ln->cond.p, cond + (size_t)ln->b0 * (ln->n_cond / ln->nb), ln->n_cond * 4,
                                          hipMemcpyDeviceToDevice, st));
        if (sampler_pack_x(s, ln, st)) return 1;
        if (launch_step_counter(ln->step.as<int>(), ln->fused_tail ? -1 : 0, 0, st)) return 1;
        const float* noise = s->cfg.mode == RLDM_SAMPLER_DDPM ? step_noise + lat_off : nullptr;
        if (!ln->step_graph || ln->captured_noise != noise) {
            // one eager step first (sets kernel attributes, packs weights), then rewind and capture
            if (sampler_enqueue_step(s, ln, noise, st)) return 1;
            RLDM_HIP_CHECK(hipStreamSynchronize(st));
            if (ln->uplan->trunk_error.p) {         // the persistent trunk's self-check (a wait that gave up / a cluster off its XCD)
                int terr = 0;
                RLDM_HIP_CHECK(hipMemcpy(&terr, ln->uplan->trunk_error.p, 4, hipMemcpyDeviceToHost));
                const bool off = ((g_dbg_flags | s->plan_flags) & (1 << 24)) != 0;
                if (getenv("RLDM_TEST_TRUNK_FAIL") && !off) terr = 2;      // (tests: the fall-back path below)
                if (terr != 0 && !off) {
                    // the clusters' only assumption (an image's workgroups share an XCD; all of them resident) does not hold on this
                    // device / driver / right now: THIS sampler runs every layer as a launch of its own from here on (same kernels, tiles)
                    if (sampler_drop_persistent(s, "persistent launches failed their self-check at the warm-up step", terr)) return 1;
                    return rldm_sample(s, x_T, step_noise, cond, images, latents_out, stream);
                }
                RLDM_REQUIRE(terr == 0, "persistent trunk launch failed its self-check (code " + std::to_string(terr) +
                                            "): set RLDM_DBG_FLAGS=16777216 to run the levels as separate launches");
            }
            RLDM_HIP_CHECK(hipMemcpyAsync(ln->x.p, x_T + lat_off, ln->n_latent * 4, hipMemcpyDeviceToDevice, st));
            if (sampler_pack_x(s, ln, st)) return 1;
            if (launch_step_counter(ln->step.as<int>(), ln->fused_tail ? -1 : 0, 0, st)) return 1;
            // the graph holds `gs` consecutive steps (the step index lives on the device, so the steps are identical launches):
            // fewer, longer graphs keep the queue fed across step boundaries
            int gs = getenv("RLDM_GRAPH_STEPS") ? atoi(getenv("RLDM_GRAPH_STEPS")) : 10;
            gs = std::max(1, std::min(gs, s->cfg.num_steps));
            while (s->cfg.num_steps % gs != 0) --gs;
            ln->graph_steps = gs;
            if (capture(st, [&]() {
                    for (int k = 0; k < gs; ++k)
                        if (sampler_enqueue_step(s, ln, noise, st)) return 1;
                    return 0;
                }, &ln->step_graph)) return 1;
            ln->captured_noise = noise;
        }
        if (images && s->vae && !ln->decode_graph) {
            // (the decode runs on whatever x holds now: only its launch list is recorded)
            if (ln->dplan->run(st)) return 1;
            RLDM_HIP_CHECK(hipStreamSynchronize(st));
            if (capture(st, [&]() { return ln->dplan->run(st); }, &ln->decode_graph)) return 1;
            RLDM_HIP_CHECK(hipMemcpyAsync(ln->x.p, x_T + lat_off, ln->n_latent * 4, hipMemcpyDeviceToDevice, st));
            if (sampler_pack_x(s, ln, st)) return 1;
        }
    }
    if (s->inject_error) {                          // tests: a cluster wait that gave up in the middle of a run
        for (auto& lnp : s->lanes)
            if (lnp->uplan->trunk_error.p && launch_step_counter(lnp->uplan->trunk_error.as<int>(), s->inject_error, 0, lnp->stream)) return 1;
        s->inject_error = 0;
    }
    // the chains: step-major so every stream always has work queued
    for (int i = 0; i < s->cfg.num_steps; i += s->lanes[0]->graph_steps)
        for (auto& lnp : s->lanes) RLDM_HIP_CHECK(hipGraphLaunch(lnp->step_graph, lnp->stream));
    for (auto& lnp : s->lanes) {
        SamplerLane* ln = lnp.get();
        hipStream_t st = ln->stream;
        const size_t lat_off = (size_t)ln->b0 * (ln->n_latent / ln->nb);
        if (latents_out)
            RLDM_HIP_CHECK(hipMemcpyAsync(latents_out + lat_off, ln->x.p, ln->n_latent * 4, hipMemcpyDeviceToDevice, st));
        if (images) {
            if (s->vae) {
                RLDM_HIP_CHECK(hipGraphLaunch(ln->decode_graph, st));
                RLDM_HIP_CHECK(hipMemcpyAsync(images + (size_t)ln->b0 * (ln->n_image / ln->nb), ln->image.p, ln->n_image * 4,
                                              hipMemcpyDeviceToDevice, st));
            } else {
                RLDM_HIP_CHECK(hipMemcpyAsync(images + lat_off, ln->x.p, ln->n_latent * 4, hipMemcpyDeviceToDevice, st));
            }
        }
        if (ln->uplan->trunk_error.p) {
            // same-call failure signal: the call's outputs are NaN-marked on the device if a persistent launch gave up a wait, and
            // the word travels to pinned memory for rldm_sampler_status (no host synchronisation here)
            if (!ln->check_host) {
                RLDM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&ln->check_host), 64, hipHostMallocDefault));
                *ln->check_host = 0;
            }
            float* img_l = images ? images + (s->vae ? (size_t)ln->b0 * (ln->n_image / ln->nb) : lat_off) : nullptr;
            if (launch_trunk_check(ln->uplan->trunk_error.as<int>(), img_l, s->vae ? ln->n_image : ln->n_latent,
                                   latents_out ? latents_out + lat_off : nullptr, ln->n_latent, st)) return 1;
            RLDM_HIP_CHECK(hipMemcpyAsync(ln->check_host, ln->uplan->trunk_error.p, 4, hipMemcpyDeviceToHost, st));
            ln->check_pending = true;
        }
        RLDM_HIP_CHECK(hipEventRecord(ln->ev_out, st));
        ln->call_recorded = true;
        RLDM_HIP_CHECK(hipStreamWaitEvent(caller, ln->ev_out, 0));
    }
    return 0;
}

// one instrumented UNet step + scheduler step (+ VAE decode) of lane 0: per-kernel launch counts, HIP-event time,
// algorithmic work.  JSON also carries "lanes" and "lane_batch" so the caller can scale to the whole batch.
int rldm_sampler_profile(rldm_sampler* s, const float* x_T, char* json_out, size_t cap) {
    RLDM_REQUIRE(s && x_T && json_out && cap > 2, "null argument");
    SamplerLane* ln = s->lanes[0].get();
    hipStream_t st = ln->stream;
    RLDM_HIP_CHECK(hipMemcpyAsync(ln->x.p, x_T, ln->n_latent * 4, hipMemcpyDeviceToDevice, st));
    if (sampler_pack_x(s, ln, st)) return 1;
    if (launch_step_counter(ln->step.as<int>(), ln->fused_tail ? -1 : 0, 0, st)) return 1;
    if (ln->uplan->run(st)) return 1;                      // warm
    RLDM_HIP_CHECK(hipStreamSynchronize(st));
    std::map<std::string, KernelStat> unet_stats, vae_stats;
    if (ln->uplan->run_profiled(st, unet_stats)) return 1;
    if (ln->dplan) {
        if (ln->dplan->run(st)) return 1;
        RLDM_HIP_CHECK(hipStreamSynchronize(st));
        if (ln->dplan->run_profiled(st, vae_stats)) return 1;
    }
    std::string js = "{";
    auto dump = [&](const char* key, const std::map<std::string, KernelStat>& m) {
        js += std::string("\"") + key + "\": {";
        bool first = true;
        for (auto& kv : m) {
            if (!first) js += ", ";
            first = false;
            char buf[512];
            snprintf(buf, sizeof(buf), "\"%s\": {\"launches\": %d, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
                     kv.first.c_str(), kv.second.launches, kv.second.ms, kv.second.flops, kv.second.bytes);
            js += buf;
        }
        js += "}";
    };
    dump("unet_step", unet_stats);
    js += ", ";
    dump("vae_decode", vae_stats);
    // routing bits in effect (rldm_sampler_config::plan_flags | the library's own fall-backs) and whether a fall-back engaged
    js += ", \"plan_flags\": " + std::to_string(s->plan_flags) + ", \"fell_back\": " + (s->plan_flags != s->cfg.plan_flags ? "true" : "false");
    js += ", \"lanes\": " + std::to_string(s->lanes.size()) + ", \"lane_batch\": " + std::to_string(ln->nb) + "}";
    RLDM_REQUIRE(js.size() + 1 <= cap, "profile buffer too small");
    memcpy(json_out, js.c_str(), js.size() + 1);
    return 0;
}

// ---- kernel-level test entry points ---------------------------------------------------------------------------------
namespace {
struct ConvCase {
    Layers layers;
    Plan plan;
    Tensor t0, t1, tr, out;
    int Wout = 0, Hout = 0, C0p = 0;
};
// one fused conv as a plan: [statistics of x0/x1 when gn] + conv
int make_conv_case(ConvCase& cc, const rldm_conv_desc* d, const float* weight, const float* bias, const float* gamma,
                   const float* beta, int res_channels, bool with_temb) {
    const bool with_res = res_channels > 0;
    const int Cin = d->Cin0 + d->Cin1;
    cc.C0p = d->Cin1 ? d->Cin0 : pad16(d->Cin0);
    RLDM_REQUIRE(d->Cin1 == 0 || (d->Cin0 % 16 == 0 && Cin % 16 == 0), "concat test needs Cin0 % 16 == 0 and Cin % 16 == 0");
    ParamStore ps;
    ps.host["c.weight"].assign(weight, weight + (size_t)d->Cout * Cin * d->ksize * d->ksize);
    ps.host["c.bias"].assign(bias, bias + d->Cout);
    if (cc.layers.add_conv(ps, "c", d->Cout, Cin, d->ksize)) return 1;
    if (with_res) {                                   // `+ res`: identity residual phase over Cout channels, or (bench
        ConvLayer* L = cc.layers.get_conv("c");       // only) a synthetic 1x1 shortcut over res_channels
        L->R = res_channels;
        L->sc_identity = res_channels == d->Cout;
        if (!L->sc_identity) {
            L->sc_w.resize((size_t)d->Cout * res_channels);
            uint32_t rng = 777u;
            for (auto& v : L->sc_w) {
                rng = rng * 1664525u + 1013904223u;
                v = (((rng >> 8) & 0xffff) / 32768.0f - 1.0f) / std::sqrt((float)res_channels);
            }
        }
    }
    if (d->gn) {
        RLDM_REQUIRE(gamma && beta && cc.C0p + d->Cin1 == Cin, "GroupNorm test needs unpadded channels");
        ps.host["n.weight"].assign(gamma, gamma + Cin);
        ps.host["n.bias"].assign(beta, beta + Cin);
        if (cc.layers.add_norm(ps, "n", Cin)) return 1;
    }
    const int s = d->stride, up = d->upsample ? 2 : 1;
    cc.Wout = d->Win * up / s;
    cc.Hout = d->Hin * up / s;
    auto walk = [&cc, d, with_res, res_channels, with_temb, s, up](Builder& bb) -> int {
        cc.t0 = bb.make(d->B, d->Win, d->Hin, cc.C0p);
        cc.t1 = Tensor();
        cc.tr = Tensor();
        if (d->Cin1) cc.t1 = bb.make(d->B, d->Win, d->Hin, d->Cin1);
        if (with_res) cc.tr = bb.make(d->B, cc.Wout, cc.Hout, res_channels);
        if (d->gn) {
            if (bb.gn_stats(cc.t0)) return 1;
            if (d->Cin1 && bb.gn_stats(cc.t1)) return 1;
        }
        ConvArgs a;
        a.layer = cc.layers.get_conv("c");
        a.x0 = cc.t0; a.x1 = cc.t1;
        a.stride = s; a.pad_mode = d->pad_mode; a.up = up;
        a.gn = d->gn ? cc.layers.get_norm("n") : nullptr;
        a.eps = d->eps; a.silu = d->silu;
        a.temb_off = with_temb ? 0 : -1;
        a.r0 = cc.tr;
        a.want_stats = true;
        // rldm_debug_set_flags2(1 << 26): a conv of <= 4 output channels is a network OUTPUT layer -- fp32 NCHW into plan.io.out (the VAE
        // decoder's conv_out), no bf16 tensor, no statistics
        if ((dbg2() & (1 << 26)) && d->Cout <= 4 && !with_res && !with_temb) {
            a.out_f32_nchw = true;
            a.want_stats = false;
        }
        return bb.conv(a, &cc.out);
    };
    return build_plan(&cc.plan, d->Cout, walk);
}
}  // namespace

int rldm_test_conv(const rldm_conv_desc* d, const float* x0, const float* x1, const float* weight, const float* bias,
                   const float* gamma, const float* beta, const float* temb, const float* res, float* y, void* stream) {
    RLDM_REQUIRE(d && x0 && weight && bias && y, "null argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ConvCase cc;
    if (make_conv_case(cc, d, weight, bias, gamma, beta, res ? d->Cout : 0, temb != nullptr)) return 1;
    DevBuf tembd;
    if (temb && upload(tembd, temb, (size_t)d->B * d->Cout * 4)) return 1;
    char* base = cc.plan.arena.as<char>();
    auto tp = [&](const Tensor& t) { return reinterpret_cast<bf16_t*>(base + t.off); };
    if (launch_nchw_f32_to_nhwc_bf16(x0, tp(cc.t0), d->B, d->Cin0, d->Win, d->Hin, cc.C0p, st)) return 1;
    if (d->Cin1 && launch_nchw_f32_to_nhwc_bf16(x1, tp(cc.t1), d->B, d->Cin1, d->Win, d->Hin, d->Cin1, st)) return 1;
    if (res && launch_nchw_f32_to_nhwc_bf16(res, tp(cc.tr), d->B, d->Cout, cc.Wout, cc.Hout, d->Cout, st)) return 1;
    cc.plan.io.temb = tembd.as<float>();
    cc.plan.io.temb_rows_per_step = d->B;
    cc.plan.io.temb_per_sample = 1;
    if (!cc.out.valid()) cc.plan.io.out = y;         // (an output layer: fp32 NCHW straight into the caller's buffer)
    if (cc.plan.run(st)) return 1;
    if (cc.out.valid() && launch_nhwc_bf16_to_nchw_f32(tp(cc.out), y, d->B, d->Cout, cc.Wout, cc.Hout, d->Cout, st)) return 1;
    RLDM_HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}

// per-channel statistics the conv epilogue emitted for its output: (sum, sumsq) over each image -> stats [B][Cout][2]
int rldm_test_conv_stats(const rldm_conv_desc* d, const float* x0, const float* weight, const float* bias, float* stats,
                         void* stream) {
    RLDM_REQUIRE(d && x0 && weight && bias && stats, "null argument");
    RLDM_REQUIRE(d->Cin1 == 0 && !d->gn, "stats test: plain single-input conv only");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ConvCase cc;
    if (make_conv_case(cc, d, weight, bias, nullptr, nullptr, 0, false)) return 1;
    char* base = cc.plan.arena.as<char>();
    if (launch_nchw_f32_to_nhwc_bf16(x0, reinterpret_cast<bf16_t*>(base + cc.t0.off), d->B, d->Cin0, d->Win, d->Hin, cc.C0p, st)) return 1;
    if (cc.plan.run(st)) return 1;
    RLDM_HIP_CHECK(hipStreamSynchronize(st));
    RLDM_REQUIRE(cc.out.P > 0, "internal: conv output carries no statistics");
    std::vector<float2> part((size_t)d->B * cc.out.P * d->Cout);
    RLDM_HIP_CHECK(hipMemcpy(part.data(), base + cc.out.st_off, part.size() * sizeof(float2), hipMemcpyDeviceToHost));
    std::vector<float> hs((size_t)d->B * d->Cout * 2, 0.f);
    for (int b = 0; b < d->B; ++b)
        for (int q = 0; q < cc.out.P; ++q)
            for (int c = 0; c < d->Cout; ++c) {
                const float2 v = part[((size_t)b * cc.out.P + q) * d->Cout + c];
                hs[((size_t)b * d->Cout + c) * 2] += v.x;
                hs[((size_t)b * d->Cout + c) * 2 + 1] += v.y;
            }
    RLDM_HIP_CHECK(hipMemcpy(stats, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    return 0;
}

int rldm_debug_set_flags2(int flags) {
    g_dbg_flags2 = flags;
    return 0;
}
int rldm_debug_set_flags(int flags) {
    g_dbg_flags = flags;
    return 0;
}

// enable (host_out == NULL: allocate + zero) or read back (host_out = 256 x u64) the in-kernel timestamps of the conv kernel
static constexpr int kTsBlocks = 2048;          // per-workgroup [start, end] s_memrealtime pairs behind the 256 stamps
int rldm_debug_timestamps(unsigned long long* host_out) {
    if (!g_ts_buf) {
        RLDM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&g_ts_buf), (256 + 2 * kTsBlocks) * 8));
        RLDM_HIP_CHECK(hipMemset(g_ts_buf, 0, (256 + 2 * kTsBlocks) * 8));
    }
    if (host_out) {
        RLDM_HIP_CHECK(hipDeviceSynchronize());
        RLDM_HIP_CHECK(hipMemcpy(host_out, g_ts_buf, 256 * 8, hipMemcpyDeviceToHost));
        RLDM_HIP_CHECK(hipMemset(g_ts_buf, 0, 256 * 8));
    }
    return 0;
}
// ABLATE builds of conv_stream.hip: [start, end] of every workgroup (first 2048) of the last launch on the 100 MHz
// s_memrealtime counter, which all XCDs share -- the spread of the starts / ends is the launch's dispatch and tail skew
int rldm_debug_block_times(unsigned long long* host_out, int nblocks) {
    RLDM_REQUIRE(host_out && nblocks >= 1 && nblocks <= kTsBlocks, "block times: 1..2048 blocks");
    RLDM_REQUIRE(g_ts_buf != nullptr, "block times: call rldm_debug_timestamps(NULL) first");
    RLDM_HIP_CHECK(hipDeviceSynchronize());
    RLDM_HIP_CHECK(hipMemcpy(host_out, g_ts_buf + 256, (size_t)nblocks * 16, hipMemcpyDeviceToHost));
    RLDM_HIP_CHECK(hipMemset(g_ts_buf + 256, 0, 2 * kTsBlocks * 8));
    return 0;
}

// timeline of the UNet ops inside the sampler's captured step graph (flag 8192 at sampler creation): stamps[i] =
// 100 MHz counter before op i (stamps[n] after the last), names = op kernel names joined by '\n'.  Returns the op count.
int rldm_debug_graph_trace(unsigned long long* stamps, int cap, char* names, size_t names_cap) {
    if (!g_trace.p || !g_trace_plan) return 0;
    (void)hipDeviceSynchronize();
    int n = 0;                                  // (the ops that launched: run_stamped leaves no stamp for the others)
    for (auto& o : g_trace_plan->ops) n += (!o.active || o.active()) ? 1 : 0;
    const int m = std::min(cap, n + 1);
    if (stamps && m > 0 && hipMemcpy(stamps, g_trace.p, (size_t)m * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (names && names_cap) {
        std::string all;
        for (auto& o : g_trace_plan->ops)
            if (!o.active || o.active()) all += o.name + "\n";
        strncpy(names, all.c_str(), names_cap - 1);
        names[names_cap - 1] = 0;
    }
    return n;
}

int rldm_debug_force_tile(int BM, int BN, int ksplit) {
    g_force_bm = BM;
    g_force_bn = BN;
    g_force_ks = ksplit;
    return 0;
}

// times the fused conv kernel alone (HIP events on `stream`) on synthetic device data; the statistics launches of a
// GroupNorm case run once before the timed region
int rldm_bench_conv(const rldm_conv_desc* d, int with_res, int with_temb, int warmup, int iters, float* avg_us,
                    char* kernel_name, size_t name_cap, void* stream) {
    RLDM_REQUIRE(d && avg_us && iters >= 1, "bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int Cin = d->Cin0 + d->Cin1;
    const int taps = d->ksize * d->ksize;
    std::vector<float> w((size_t)d->Cout * Cin * taps), bias(d->Cout), gamma(Cin, 1.f), beta(Cin, 0.f);
    uint32_t rng = 12345u;
    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xffff) / 32768.0f - 1.0f; };
    const float ws = 1.0f / std::sqrt((float)Cin * taps);
    for (auto& v : w) v = rnd() * ws;
    for (auto& v : bias) v = rnd() * 0.1f;
    ConvCase cc;
    if (make_conv_case(cc, d, w.data(), bias.data(), gamma.data(), beta.data(), with_res, with_temb != 0)) return 1;
    DevBuf tembd;
    if (with_temb) {
        std::vector<float> t((size_t)d->B * d->Cout);
        for (auto& v : t) v = rnd();
        if (upload(tembd, t.data(), t.size() * 4)) return 1;
    }
    {   // synthetic activations: bf16 uniform(-1, 1)
        std::vector<bf16_t> h(cc.plan.arena.bytes / 2);
        for (auto& v : h) v = f32_to_bf16(rnd());
        RLDM_HIP_CHECK(hipMemcpy(cc.plan.arena.p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    }
    cc.plan.io.temb = tembd.as<float>();
    cc.plan.io.temb_rows_per_step = d->B;
    cc.plan.io.temb_per_sample = 1;
    DevBuf outd;
    if (!cc.out.valid()) {
        if (outd.alloc((size_t)d->B * d->Cout * cc.Wout * cc.Hout * sizeof(float))) return 1;
        cc.plan.io.out = outd.as<float>();
    }
    RLDM_REQUIRE(!cc.plan.ops.empty(), "internal: empty plan");
    // the timed unit: the conv launch, plus the GroupNorm+SiLU launch in front of it on the conv_small.hip route
    size_t last = cc.plan.ops.size() - 1;           // the conv itself: the last op whose name starts with "conv_" (a gn_fold of
    while (last > 0 && cc.plan.ops[last].name.rfind("conv_", 0) != 0) --last;      // its output statistics may follow it)
    size_t first = last;
    if (first > 0 && cc.plan.ops[first - 1].name == "gn_apply_kernel") --first;
    auto run_unit = [&]() {
        for (size_t o = first; o <= last; ++o)
            if (cc.plan.ops[o].fn(st)) return 1;
        return 0;
    };
    if (kernel_name && name_cap) {
        const std::string nm = (first < last ? "gn_apply+" : "") + cc.plan.ops[last].name;
        strncpy(kernel_name, nm.c_str(), name_cap - 1);
        kernel_name[name_cap - 1] = 0;
    }
    if (cc.plan.run(st)) return 1;
    for (int i = 0; i < warmup; ++i)
        if (run_unit()) return 1;
    hipEvent_t e0, e1;
    RLDM_HIP_CHECK(hipEventCreate(&e0));
    RLDM_HIP_CHECK(hipEventCreate(&e1));
    // RLDM_BENCH_THRASH_MB=n: a sweep over n MB in front of every timed launch (not timed itself): the conv then finds its weights
    // and input out of the L2s (n ~ 96: Infinity-Cache warm; n >= 512: HBM cold) -- the states a layer meets inside a sampler step
    const size_t thrash = getenv("RLDM_BENCH_THRASH_MB") ? (size_t)atoi(getenv("RLDM_BENCH_THRASH_MB")) << 20 : 0;
    if (thrash) {
        DevBuf tb, sink;
        if (tb.alloc(thrash) || sink.alloc(64)) return 1;
        RLDM_HIP_CHECK(hipMemsetAsync(tb.p, 0, thrash, st));
        double tot = 0.0;
        for (int i = 0; i < iters; ++i) {
            if (launch_thrash(tb.p, thrash, sink.as<float>(), st)) return 1;
            RLDM_HIP_CHECK(hipEventRecord(e0, st));
            if (run_unit()) return 1;
            RLDM_HIP_CHECK(hipEventRecord(e1, st));
            RLDM_HIP_CHECK(hipStreamSynchronize(st));
            float ms = 0.f;
            RLDM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            tot += ms;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *avg_us = (float)(tot * 1000.0 / iters);
        return 0;
    }
    RLDM_HIP_CHECK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i)
        if (run_unit()) return 1;
    RLDM_HIP_CHECK(hipEventRecord(e1, st));
    RLDM_HIP_CHECK(hipStreamSynchronize(st));
    float ms = 0.f;
    RLDM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_us = ms * 1000.0f / iters;
    return 0;
}

int rldm_test_attention(const float* qkv, int B, int L, int C, float* out, void* stream) {
    RLDM_REQUIRE(qkv && out, "null argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // host-side conversion keeps this test path trivial: qkv [B][L][3C] bf16 with q pre-scaled by log2(e)/sqrt(8)
    std::vector<float> h((size_t)B * L * 3 * C);
    RLDM_HIP_CHECK(hipMemcpy(h.data(), qkv, h.size() * 4, hipMemcpyDeviceToHost));
    std::vector<bf16_t> hq(h.size());
    const float qs = 1.4426950408889634f / std::sqrt(8.0f);
    for (size_t i = 0; i < h.size(); ++i) hq[i] = f32_to_bf16((int)(i % (3 * (size_t)C)) < C ? h[i] * qs : h[i]);
    DevBuf dq, dout;
    if (upload(dq, hq.data(), hq.size() * 2)) return 1;
    if (dout.alloc((size_t)B * L * C * 2)) return 1;
    AttnParams ap;
    ap.qkv = dq.as<bf16_t>(); ap.out = dout.as<bf16_t>();
    ap.B = B; ap.L = L; ap.C = C;
    if (launch_attention(ap, st)) return 1;
    std::vector<bf16_t> ho((size_t)B * L * C);
    RLDM_HIP_CHECK(hipStreamSynchronize(st));
    RLDM_HIP_CHECK(hipMemcpy(ho.data(), dout.p, ho.size() * 2, hipMemcpyDeviceToHost));
    std::vector<float> hf(ho.size());
    for (size_t i = 0; i < ho.size(); ++i) hf[i] = bf16_to_f32(ho[i]);
    RLDM_HIP_CHECK(hipMemcpy(out, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
    return 0;
}

// The fused GroupNorm -> to_q/to_k/to_v -> softmax(q k^T / sqrt(8)) v launch on its own (what runs inside every
// attention block of the UNet): x device fp32 [B][L][C] (token-major), gamma/beta host [C], wqkv host [3C][C] (rows
// q | k | v, torch Linear layout), bqkv host [3C] -> out device fp32 [B][L][C] (heads concatenated, before to_out).
namespace {
struct AttnCase {
    DevBuf dx, dst, dg, dbt, dw, dbias, dout;
    AttnQkvParams ap;
};
// x_bf16 host [B][L][C]; builds the per-image statistics row, the per-head fragments and the launch parameters
int make_attn_case(AttnCase& ac, const std::vector<bf16_t>& hb, int B, int L, int C, int groups, float eps, const float* gamma,
                   const float* beta, const float* wqkv, const float* bqkv) {
    const size_t n = (size_t)B * L * C;
    std::vector<float> stats((size_t)B * C * 2, 0.f);      // one partial row per image: (sum, sum of squares) of the bf16 values
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            double S = 0.0, SS = 0.0;
            for (int l = 0; l < L; ++l) {
                const double v = bf16_to_f32(hb[((size_t)b * L + l) * C + c]);
                S += v;
                SS += v * v;
            }
            stats[((size_t)b * C + c) * 2] = (float)S;
            stats[((size_t)b * C + c) * 2 + 1] = (float)SS;
        }
    std::vector<float> w((size_t)3 * C * C), bb((size_t)3 * C);
    const float qs = 1.4426950408889634f / std::sqrt(8.0f);
    for (size_t i = 0; i < w.size(); ++i) w[i] = wqkv[i] * (i < (size_t)C * C ? qs : 1.f);
    for (int i = 0; i < 3 * C; ++i) bb[i] = bqkv[i] * (i < C ? qs : 1.f);
    std::vector<bf16_t> img;
    std::vector<float> bias;
    pack_attn_head_frags(w.data(), bb.data(), C, img, bias);
    if (upload(ac.dx, hb.data(), n * 2) || upload(ac.dst, stats.data(), stats.size() * 4) || upload(ac.dg, gamma, C * 4) ||
        upload(ac.dbt, beta, C * 4) || upload(ac.dw, img.data(), img.size() * 2) || upload(ac.dbias, bias.data(), bias.size() * 4))
        return 1;
    if (ac.dout.alloc(n * 2)) return 1;
    AttnQkvParams& ap = ac.ap;
    memset(&ap, 0, sizeof(ap));
    ap.x = ac.dx.as<bf16_t>(); ap.st = ac.dst.as<float2>(); ap.P = 1;
    ap.gamma = ac.dg.as<float>(); ap.beta = ac.dbt.as<float>(); ap.eps = eps; ap.groups = groups;
    const int cpg = C / groups;
    ap.inv_n = (float)(1.0 / ((double)L * cpg));
    ap.magic_cpg = ((1 << 20) + cpg - 1) / cpg;
    ap.wfrag = ac.dw.as<bf16_t>(); ap.bias = ac.dbias.as<float>(); ap.out = ac.dout.as<bf16_t>();
    ap.B = B; ap.L = L; ap.C = C;
    ap.ts = getenv("RLDM_TS_TRUNK") ? nullptr : g_ts_buf;                                  // ABLATE builds: phase stamps (rldm_debug_timestamps)
    ap.ts_L = L;
    return 0;
}
}  // namespace

int rldm_test_attention_qkv(const float* x, int B, int L, int C, int groups, float eps, const float* gamma, const float* beta,
                            const float* wqkv, const float* bqkv, float* out, void* stream) {
    RLDM_REQUIRE(x && gamma && beta && wqkv && bqkv && out, "null argument");
    RLDM_REQUIRE(C % 16 == 0 && C % groups == 0, "attention_qkv: channels must be a multiple of 16 and of the group count");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t n = (size_t)B * L * C;
    std::vector<float> hx(n);
    RLDM_HIP_CHECK(hipMemcpy(hx.data(), x, n * 4, hipMemcpyDeviceToHost));
    std::vector<bf16_t> hb(n);
    for (size_t i = 0; i < n; ++i) hb[i] = f32_to_bf16(hx[i]);
    AttnCase ac;
    if (make_attn_case(ac, hb, B, L, C, groups, eps, gamma, beta, wqkv, bqkv)) return 1;
    if (launch_attention_qkv(ac.ap, st)) return 1;
    RLDM_HIP_CHECK(hipStreamSynchronize(st));
    std::vector<bf16_t> ho(n);
    RLDM_HIP_CHECK(hipMemcpy(ho.data(), ac.dout.p, n * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) hx[i] = bf16_to_f32(ho[i]);
    RLDM_HIP_CHECK(hipMemcpy(out, hx.data(), n * 4, hipMemcpyHostToDevice));
    return 0;
}

// times the fused attention launch alone (HIP events on `stream`) on synthetic data of the given geometry
int rldm_bench_attention_qkv(int B, int L, int C, int warmup, int iters, float* avg_us, void* stream) {
    RLDM_REQUIRE(avg_us && iters >= 1 && C % 32 == 0, "bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t n = (size_t)B * L * C;
    uint32_t rng = 4242u;
    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xffff) / 32768.0f - 1.0f; };
    std::vector<bf16_t> hb(n);
    for (auto& v : hb) v = f32_to_bf16(rnd() * 1.7f + 0.2f);
    std::vector<float> gamma(C, 1.f), beta(C, 0.f), w((size_t)3 * C * C), bb((size_t)3 * C, 0.f);
    const float ws = 2.0f / std::sqrt((float)C);
    for (auto& v : w) v = rnd() * ws;
    AttnCase ac;
    if (make_attn_case(ac, hb, B, L, C, 32, 1e-5f, gamma.data(), beta.data(), w.data(), bb.data())) return 1;
    for (int i = 0; i < warmup; ++i)
        if (launch_attention_qkv(ac.ap, st)) return 1;
    hipEvent_t e0, e1;
    RLDM_HIP_CHECK(hipEventCreate(&e0));
    RLDM_HIP_CHECK(hipEventCreate(&e1));
    RLDM_HIP_CHECK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i)
        if (launch_attention_qkv(ac.ap, st)) return 1;
    RLDM_HIP_CHECK(hipEventRecord(e1, st));
    RLDM_HIP_CHECK(hipStreamSynchronize(st));
    float ms = 0.f;
    RLDM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_us = ms * 1000.0f / iters;
    return 0;
}

}  // extern "C"
