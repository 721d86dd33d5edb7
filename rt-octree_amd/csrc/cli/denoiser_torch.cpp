#include "denoiser_torch.h"

#include <torch/script.h>

#include "rto.h"

#include <iostream>
#include <stdexcept>

namespace rto {

struct TorchDenoiser::Impl {
    torch::jit::script::Module module;
    torch::Tensor weight, guidance;
    int device = 0;
    rto_guidance_net* fused = nullptr;  // non-null: forward runs librto's fused kernel
    int fused_levels = 0;
};

TorchDenoiser::TorchDenoiser(const std::string& path, int device, bool use_fused) : impl_(new Impl) {
    if (path.empty()) throw std::runtime_error("No torchscript module is given to denoiser.");  // denoiser.cpp:13-16
    impl_->device = device;
    try {
        impl_->module = torch::jit::load(path, torch::Device(torch::kCUDA, (c10::DeviceIndex)device));
        impl_->module.eval();
    } catch (const c10::Error& e) {  // denoiser.cpp:22-26
        std::cerr << e.what() << std::endl;
        throw std::runtime_error("Error when loading torchscript model from " + path);
    }
    if (use_fused) {
        torch::Tensor w1, b1, w2, b2;
        int n_params = 0;
        for (const auto& p : impl_->module.named_parameters()) {
            ++n_params;
            if (p.name == "layers.0.conv.weight") w1 = p.value;
            if (p.name == "layers.0.conv.bias") b1 = p.value;
            if (p.name == "layers.1.conv.weight") w2 = p.value;
            if (p.name == "layers.1.conv.bias") b2 = p.value;
        }
        if (n_params == 4 && w1.defined() && b1.defined() && w2.defined() && b2.defined() && w1.dim() == 4 &&
            w2.dim() == 4 && w1.size(1) == 8 && w1.size(2) == 3 && w1.size(3) == 3 && w2.size(1) == w1.size(0) &&
            w2.size(2) == 3 && w2.size(3) == 3) {
            auto host = [](const torch::Tensor& t) { return t.detach().to(torch::kCPU, torch::kFloat32).contiguous(); };
            const torch::Tensor hw1 = host(w1), hb1 = host(b1), hw2 = host(w2), hb2 = host(b2);
            rto_guidance_net* net = nullptr;
            const int c1 = (int)w1.size(0), levels = (int)w2.size(0) / 2;
            if (rto_guidance_net_create(hw1.data_ptr<float>(), hb1.data_ptr<float>(), hw2.data_ptr<float>(),
                                        hb2.data_ptr<float>(), c1, levels, device, &net) == RTO_OK) {
                impl_->fused = net;  // (RTO_E_UNSUPPORTED for other widths: stay on libtorch)
                impl_->fused_levels = levels;
            }
        }
    }
}

TorchDenoiser::~TorchDenoiser() {
    if (impl_ && impl_->fused) rto_guidance_net_free(impl_->fused);
}

bool TorchDenoiser::fused() const { return impl_->fused != nullptr; }

void TorchDenoiser::forward(float* aux, int n, int H, int W, const float** weight, const float** guidance, int* levels) {
    torch::NoGradGuard no_grad;
    const auto options = torch::TensorOptions().device(torch::kCUDA, impl_->device).dtype(torch::kFloat32);
    if (impl_->fused) {  // same maps from one HIP kernel on the default stream (the CLI's stream)
        const int L = impl_->fused_levels;
        if (!impl_->weight.defined() || impl_->weight.size(0) != n || impl_->weight.size(2) != H || impl_->weight.size(3) != W) {
            impl_->weight = torch::empty({n, L, H, W}, options);
            impl_->guidance = torch::empty({n, L, H, W}, options);
        }
        if (rto_guidance_net_forward(impl_->fused, nullptr, aux, n, H, W, impl_->weight.data_ptr<float>(),
                                     impl_->guidance.data_ptr<float>()) != RTO_OK)
            throw std::runtime_error(std::string("fused GuidanceNet failed: ") + rto_last_error());
        *weight = impl_->weight.data_ptr<float>();
        *guidance = impl_->guidance.data_ptr<float>();
        *levels = L;
        return;
    }
    torch::Tensor aux_t = torch::from_blob(aux, {n, 8, H, W}, options);  // denoiser.cpp:40-43 (n = 1 there)
    auto maps = impl_->module.forward({aux_t}).toTuple()->elements();
    impl_->weight = maps[0].toTensor().contiguous();    // [n,L,H,W]
    impl_->guidance = maps[1].toTensor().contiguous();  // [n,L,H,W]
    *weight = impl_->weight.data_ptr<float>();
    *guidance = impl_->guidance.data_ptr<float>();
    *levels = (int)impl_->guidance.size(1);
}

}  // namespace rto
