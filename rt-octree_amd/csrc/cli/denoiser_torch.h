// denoiser_torch.h -- the libtorch side of volrend::Denoiser (renderer/src/denoiser/denoiser.cpp:9-61):
// load the TorchScript GuidanceNet (`ts_*.ts`), wrap the aux buffer, run forward -> (weight_map,
// guidance_map).  Kept behind a torch-free interface so volrend_headless.cpp itself compiles without
// the libtorch headers.
#pragma once
#include <memory>
#include <string>

struct rto_guidance_net;

namespace rto {

class TorchDenoiser {
public:
    // throws std::runtime_error("No torchscript module is given to denoiser.") on an empty path and
    // "Error when loading torchscript model from <path>" when the file cannot be loaded
    // use_fused: when the module is the two-layer compact GuidanceNet that compact_and_compile exports
    // (parameters layers.{0,1}.conv.{weight,bias}, 8 -> 32 -> 8 channels), run it as librto's fused
    // MFMA kernel (rto_guidance_net_*, same fp16 weights) instead of libtorch's convolutions; any
    // other module, or use_fused = false, takes the libtorch path.
    TorchDenoiser(const std::string& ts_module_path, int device, bool use_fused = true);
    ~TorchDenoiser();
    bool fused() const;
    // the fused kernel's handle (include/rto.h rto_guidance_net_*), or nullptr on the libtorch path
    rto_guidance_net* fused_handle() const;

    // aux: device pointer to [n,8,H,W] fp32 (zero-copy from_blob).  On return *weight / *guidance point
    // at contiguous device tensors [n,L,H,W] that stay alive until the next call.
    // input_rgba (fused network only): `aux` is the noisy image [n][H][W][4] = (r, g, b, alpha) a lean batched launch leaves
    // (rto_ctx_set_lean_outputs, RTO_NET_INPUT_RGBA) instead of the 8-plane aux buffer
    void forward(float* aux, int n, int H, int W, const float** weight, const float** guidance, int* levels, bool input_rgba = false);

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace rto
