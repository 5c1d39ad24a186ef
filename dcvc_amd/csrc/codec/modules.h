// Building blocks of the DCVC-UF networks on top of the HIP kernels: parameter store, weight
// preparation (folding / re-layout done once at set_param time) and the forward launch sequences.
//
// Reference for WHAT is computed: src/layers/layers.py:92-188 and the proxy classes in
// src/layers/extensions/inference/layers_proxy.{h,cpp}. The HOW differs: no tensor pool or
// per-module pre-allocation plan - modules are stateless launch sequences over caller-provided
// (pointer, leading-dimension) views plus three shared scratch planes, which keeps the working
// set of a block chain inside the 256 MiB Infinity Cache.
#pragma once

#include "kernels/ops.h"

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace dcvc {

// ---------------------------------------------------------------- host-side parameter store
struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<half_t> h;        // floating tensors, converted to fp16 (finalize_model: net.half())
    std::vector<int32_t> i;       // integer tensors (CDF tables)
    int64_t numel() const;
};

class ParamStore {
public:
    // dtype: 0 = fp16, 1 = fp32, 2 = int32; data is HOST memory
    void add(const std::string& name, const void* data, int dtype, const int64_t* dims, int ndim);
    const HostTensor& at(const std::string& name) const;
    bool has(const std::string& name) const { return m_map.count(name) != 0; }

private:
    std::map<std::string, HostTensor> m_map;
};

// ---------------------------------------------------------------- device memory
// Owns every device allocation of a codec instance; freed together.
class DeviceArena {
public:
    DeviceArena() = default;
    ~DeviceArena();
    DeviceArena(const DeviceArena&) = delete;
    DeviceArena& operator=(const DeviceArena&) = delete;
    void* alloc(size_t bytes);                       // zero-initialised, 256-B aligned
    half_t* alloc_half(size_t count) { return static_cast<half_t*>(alloc(count * sizeof(half_t))); }
    half_t* upload(const std::vector<half_t>& host);
    void release();
    size_t total_bytes() const { return m_total; }

private:
    std::vector<void*> m_ptrs;
    size_t m_total = 0;
};

// A view of an NHWC activation: `c` channels starting at p, pixel stride ld.
struct View {
    half_t* p = nullptr;
    int ld = 0;
    int c = 0;
    View() = default;
    View(half_t* p_, int ld_, int c_) : p(p_), ld(ld_), c(c_) {}
    View slice(int c0, int n) const { return View(p + c0, ld, n); }
};

// Shared scratch for the block-internal tensors (dc.0 out, depthwise out, ffn chunk out).
struct Scratch {
    half_t* t1 = nullptr;
    half_t* t2 = nullptr;
    half_t* t3 = nullptr;
    size_t elems = 0;       // capacity of each plane in fp16 elements
    // which of t1 / t2 holds dc.0's output: a block launch with its depthwise conv inside reads it from one plane and leaves the
    // NEXT block's in the other (DcbW::forward flips this with every such hand-over and resets it whenever a block computes its own dc.0)
    mutable int hand = 0;
};

// ---------------------------------------------------------------- weights
struct Conv1x1W {
    half_t* w = nullptr;    // [cout][cin]
    half_t* b = nullptr;    // [cout] or null
    int cin = 0, cout = 0;
    void load(const ParamStore& ps, DeviceArena& mem, const std::string& prefix);
};

// The 1x1 conv that closes a chain of blocks (y_prior_fusion.conv.3, y_spatial_prior.conv.3, decoder.conv2, recon_head.head
// ...: launches of their own in the reference, dmci_proxy.cpp:145-199, dmc_ld_proxy.cpp:420-593). Where the block kernel has
// the variant it runs in the NEXT slot of the chain's last block launch (kernels/dcb_nsplit8_kernel.h), else behind it.
struct FinW {
    Conv1x1W conv;
    half_t* packed = nullptr;        // dcb_nsplit_pack_fin stream (for a last block of width conv.cin) or null
    void load(const ParamStore& ps, DeviceArena& mem, const std::string& prefix);
};
struct FinCall {
    const FinW* w = nullptr;
    const half_t* q = nullptr;       // conv1x1_bias_with_quant's scale or null
    half_t* y = nullptr; int ldy = 0;
    bool keep_block_output = false;  // false: where the conv runs inside the block launch, the block's own output is not stored
    FinCall() = default;
    FinCall(const FinW& w_, half_t* y_, int ldy_, const half_t* q_ = nullptr) : w(&w_), q(q_), y(y_), ldy(ldy_) {}
};

// layers.py:128-159 DepthConvBlock; layers_proxy.cpp:160-206 for the folding
struct DcbW {
    bool has_adaptor = false;
    Conv1x1W adaptor, dc0, dc3, ffn0, ffn2;
    half_t* dw = nullptr;   // [9][cdc]
    half_t* packed_main = nullptr;   // dcb_nsplit.hip weight streams (full-width blocks of width 384 / 512)
    half_t* packed_dc0 = nullptr;
    half_t* packed_adaptor = nullptr;   // dcb_pair.hip: adaptor + dc.0 in one launch, where the kernel has the shape
    int c = 0;              // block width (output channels)
    int cdc = 0;            // depthwise width (c or c/2)
    int cffn = 0;           // ffn inner width after chunk-add (c or c/2)
    void load(const ParamStore& ps, DeviceArena& mem, const std::string& prefix);
    // y may alias x only when the block has no adaptor and `shortcut` is false. `alt`: a spare
    // [pixels][c] buffer for the adaptor output; with it (or without an adaptor) and y != x the
    // half-width blocks of the inter models run as ONE launch (kernels/dcb_tail.hip reads the block
    // input of neighbouring patches, so it cannot run in place)
    // Blocks whose shape the N-split kernel has (kernels/dcb_nsplit.hip) run everything behind the depthwise conv in one
    // launch, which can also compute dc.0 of the block that FOLLOWS in a chain:
    // `next` = that block (must satisfy feeds(next)), its dc.0 output then waits in s.t1 and the
    // caller passes dc0_done = true to next->forward().
    // `fin`: the conv that closes the chain this block is the last of: yfin = conv(y) - inside the block launch where the
    // kernel has the variant, as a launch of its own behind it otherwise (the caller never launches it)
    void forward(View x, View y, int H, int W, const Scratch& s, hipStream_t st, bool shortcut = false,
                 const half_t* q_fused = nullptr, const half_t* q_after = nullptr, View alt = View(),
                 const DcbW* next = nullptr, bool dc0_done = false, const FinCall* fin = nullptr) const;
    bool core_fused() const;                     // this block runs through dcb_nsplit
    // this block on an H x W grid can run as ONE launch (dcb_tail with dc.0 inside) when its input is not its output
    bool one_launch(int H, int W) const;
    bool nsplit() const { return packed_main != nullptr; }
    bool feeds(const DcbW& next) const;          // ... and can compute next's dc.0 on the way out
};

// layers.py:176-188: pixel_unshuffle(2) + 1x1 == 2x2 stride-2 conv (layers_proxy.cpp:263-264)
struct Stride2W {
    half_t* w = nullptr;    // [cout][2][2][cin]
    half_t* b = nullptr;
    int cin = 0, cout = 0;
    bool shortcut = true;   // block-level skip connection (off in the inter models' hyper / prior nets)
    DcbW block;
    void load(const ParamStore& ps, DeviceArena& mem, const std::string& prefix, bool shortcut = true);
    // x: [H][W][cin] -> tmp, y: [H/2][W/2][cout]; without the shortcut tmp may be y
    void forward(View x, View tmp, View y, int H, int W, const half_t* zeros, const Scratch& s,
                 hipStream_t st) const;
};

// layers.py:92-105 SubpelConv2x / layers_proxy.cpp:269-324. Without a bias (kernel 1) it is a
// 2x2 stride-2 transposed conv; with a bias (HT-L: kernel 1 or 3) a k x k conv + bias rounded to
// fp16 followed by pixel_shuffle(2), which needs a [H][W][4*cout] temporary.
struct SubpelW {
    half_t* w = nullptr;    // no bias: [4 = dy*2+dx][cout][cin]; biased: [4*cout][k][k][cin]
    half_t* b = nullptr;    // [4*cout] or null
    int cin = 0, cout = 0, k = 1;
    void load(const ParamStore& ps, DeviceArena& mem, const std::string& prefix);   // prefix + "conv.0.weight"
    size_t tmp_elems(int H, int W) const { return b ? static_cast<size_t>(H) * W * 4 * cout : 0; }
    // y: [2H][2W][cout]
    void forward(View x, View y, int H, int W, hipStream_t st, half_t* tmp = nullptr,
                 const half_t* zeros = nullptr) const;
};

// layers.py:162-173: SubpelConv2x + DepthConvBlock
struct UpsampleW {
    SubpelW up;
    bool shortcut = true;
    DcbW block;
    void load(const ParamStore& ps, DeviceArena& mem, const std::string& prefix, bool shortcut = true);
    // x: [H][W][cin] -> tmp, y: [2H][2W][cout]; without the shortcut tmp may be y.
    // up_tmp / zeros: only for a biased upsampler (SubpelW::tmp_elems)
    void forward(View x, View tmp, View y, int H, int W, const Scratch& s, hipStream_t st,
                 half_t* up_tmp = nullptr, const half_t* zeros = nullptr, const DcbW* next = nullptr) const;
};

// nn.Sequential of DepthConvBlocks prefix + "0.", "1.", ... (as many as the checkpoint has)
struct DcbChain {
    std::vector<DcbW> blocks;
    void load(const ParamStore& ps, DeviceArena& mem, const std::string& prefix);
    // x -> (first block) -> tmp -> ... in place ... -> (last block) -> y; q_fused_last: scale fused
    // into the last block (DepthConvBlockProxy::forward(x, quant), layers_proxy.cpp:92-95).
    // With a second temporary the blocks ping-pong between the two instead of running in place.
    void forward(View x, View tmp, View y, int H, int W, const Scratch& s, hipStream_t st,
                 const half_t* q_fused_last = nullptr, View tmp2 = View(), const FinCall* fin = nullptr) const;
    size_t size() const { return blocks.size(); }
};

// the same launch sequence over a plain array of blocks (codecs that keep DcbW[n] members)
// `after`: the first block of the chain that runs NEXT on this chain's output (it must satisfy blocks[n-1].feeds(*after)):
// the last launch also computes its dc.0 into s.t1, and that chain is then run with first_dc0_done = true - nothing else
// may touch s.t1 in between.
void run_dcb_chain(const DcbW* blocks, int n, View x, View tmp, View y, int H, int W, const Scratch& s,
                   hipStream_t st, const half_t* q_fused_last = nullptr, View tmp2 = View(), const FinCall* fin = nullptr,
                   const DcbW* after = nullptr, bool first_dc0_done = false);

// dense k x k conv weight in tap-major layout
struct ConvKW {
    half_t* w = nullptr;    // [cout][k][k][cin]
    half_t* b = nullptr;
    int cin = 0, cout = 0, k = 0;
    void load(const ParamStore& ps, DeviceArena& mem, const std::string& prefix);
};

}  // namespace dcvc
