#include "denoiser_torch.h"

#include <torch/script.h>

#include <iostream>
#include <stdexcept>

namespace rto {

struct TorchDenoiser::Impl {
    torch::jit::script::Module module;
    torch::Tensor weight, guidance;
    int device = 0;
};

TorchDenoiser::TorchDenoiser(const std::string& path, int device) : impl_(new Impl) {
    if (path.empty()) throw std::runtime_error("No torchscript module is given to denoiser.");  // denoiser.cpp:13-16
    impl_->device = device;
    try {
        impl_->module = torch::jit::load(path, torch::Device(torch::kCUDA, (c10::DeviceIndex)device));
        impl_->module.eval();
    } catch (const c10::Error& e) {  // denoiser.cpp:22-26
        std::cerr << e.what() << std::endl;
        throw std::runtime_error("Error when loading torchscript model from " + path);
    }
}

TorchDenoiser::~TorchDenoiser() = default;

void TorchDenoiser::forward(float* aux, int n, int H, int W, const float** weight, const float** guidance, int* levels) {
    torch::NoGradGuard no_grad;
    const auto options = torch::TensorOptions().device(torch::kCUDA, impl_->device).dtype(torch::kFloat32);
    torch::Tensor aux_t = torch::from_blob(aux, {n, 8, H, W}, options);  // denoiser.cpp:40-43 (n = 1 there)
    auto maps = impl_->module.forward({aux_t}).toTuple()->elements();
    impl_->weight = maps[0].toTensor().contiguous();    // [n,L,H,W]
    impl_->guidance = maps[1].toTensor().contiguous();  // [n,L,H,W]
    *weight = impl_->weight.data_ptr<float>();
    *guidance = impl_->guidance.data_ptr<float>();
    *levels = (int)impl_->guidance.size(1);
}

}  // namespace rto
