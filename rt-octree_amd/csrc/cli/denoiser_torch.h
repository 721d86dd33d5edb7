// denoiser_torch.h -- the libtorch side of volrend::Denoiser (renderer/src/denoiser/denoiser.cpp:9-61):
// load the TorchScript GuidanceNet (`ts_*.ts`), wrap the aux buffer, run forward -> (weight_map,
// guidance_map).  Kept behind a torch-free interface so volrend_headless.cpp itself compiles without
// the libtorch headers.
#pragma once
#include <memory>
#include <string>

namespace rto {

class TorchDenoiser {
public:
    // throws std::runtime_error("No torchscript module is given to denoiser.") on an empty path and
    // "Error when loading torchscript model from <path>" when the file cannot be loaded
    TorchDenoiser(const std::string& ts_module_path, int device);
    ~TorchDenoiser();

    // aux: device pointer to [n,8,H,W] fp32 (zero-copy from_blob).  On return *weight / *guidance point
    // at contiguous device tensors [n,L,H,W] that stay alive until the next call.
    void forward(float* aux, int n, int H, int W, const float** weight, const float** guidance, int* levels);

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace rto
