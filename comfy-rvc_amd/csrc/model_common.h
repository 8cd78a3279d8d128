// Host-side helpers shared by the three network graphs (hubert / rmvpe / synth).
#pragma once
#include "rvc_internal.h"
#include "ops.h"
#include <cmath>
#include <memory>

namespace rvc {

struct HostTensor { std::vector<float> data; std::vector<long long> shape; size_t numel() const { return data.size(); } };

struct TensorStore {
  std::map<std::string, HostTensor> t;
  void set(const std::string& name, const float* d, const long long* shape, int ndim) {
    HostTensor h; size_t n = 1;
    for (int i = 0; i < ndim; ++i) { h.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    h.data.assign(d, d + n);
    t[name] = std::move(h);
  }
  bool has(const std::string& name) const { return t.count(name) != 0; }
  const HostTensor& get(const std::string& name) const {
    auto it = t.find(name);
    if (it == t.end()) throw Error("missing tensor '" + name + "'");
    return it->second;
  }
  const HostTensor& get(const std::string& name, std::initializer_list<long long> shape) const {
    const HostTensor& h = get(name);
    std::vector<long long> want(shape);
    if (h.shape != want) {
      std::string s = "tensor '" + name + "' has shape [";
      for (auto v : h.shape) s += std::to_string(v) + ",";
      s += "] expected [";
      for (auto v : want) s += std::to_string(v) + ",";
      throw Error(s + "]");
    }
    return h;
  }
  void clear() { t.clear(); }
};

struct Ctx {
  int device = 0;
  size_t workspace_bytes = 0;   // sum of the model arenas (informational)
  int precision = -1;           // conv arithmetic of the models built on this context: 0 fp32 MFMA only, 1 / 2 bf16x3 split where eligible,
                                // -1 (default): the mode of the thread that finalizes the model (rvc_set_conv_precision, default 1)
};

// device vector owned by a model
struct DevVec {
  float* p = nullptr; size_t n = 0;
  void upload(const std::vector<float>& h) { dev_free(p); p = dev_upload(h.data(), h.size()); n = h.size(); }
  void upload(const float* h, size_t cnt) { dev_free(p); p = dev_upload(h, cnt); n = cnt; }
  void free_() { dev_free(p); p = nullptr; n = 0; }
};

inline std::vector<float> transpose2d(const float* w, int R, int C) {   // [R][C] -> [C][R]
  std::vector<float> o((size_t)R * C);
  for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) o[(size_t)c * R + r] = w[(size_t)r * C + c];
  return o;
}

// weight_norm(dim=0): w = v * g / ||v||  with the norm over every dim but 0
inline std::vector<float> weight_norm0(const HostTensor& v, const HostTensor& g) {
  const size_t rows = (size_t)v.shape[0], per = v.numel() / rows;
  std::vector<float> w(v.numel());
  for (size_t r = 0; r < rows; ++r) {
    double s = 0.0;
    for (size_t i = 0; i < per; ++i) { const double x = v.data[r * per + i]; s += x * x; }
    const float sc = (float)((double)g.data[r] / std::sqrt(s));
    for (size_t i = 0; i < per; ++i) w[r * per + i] = v.data[r * per + i] * sc;
  }
  return w;
}

}  // namespace rvc
