// C wrapper around the reference's own LibTorch extractor code (runtime/extractor/torch_asv_extractor.cc, compiled in
// place by oracle/Makefile.ref into oracle/_ref/libextractor_ref.so - TEST INFRASTRUCTURE, build container only).
// It exposes TorchAsvExtractor::ComputeVadEnergy (torch_asv_extractor.cc:14-62), the energy VAD the runtime applies in
// front of extract_embedding_whole, so that the numpy restatement and the HIP front-end's VAD can be pinned to outputs of
// the reference itself.  glog / gflags / yaml-cpp are replaced by the parse-only stand-ins in oracle/ref_stubs/.
#include <cstring>
#include <memory>

#include "extractor/torch_asv_extractor.h"

extern "C" {

// feats: [T][dim] row-major (column 0 = log energy); voiced: T floats (1 = voiced).  Returns 0, or -1 when the reference
// rejects the options / input (CHECK / LOG(FATAL)).
int extractor_ref_vad_energy(const float *feats, int T, int dim, float energy_threshold, float energy_mean_scale, int frames_context,
                             float proportion_threshold, float *voiced) {
  try {
    subtools::ExtractOptions eo;
    subtools::TorchAsvExtractor ex(nullptr, nullptr, eo);       // ComputeVadEnergy touches neither the pipeline nor the model
    subtools::VadEnergyOptions vo;
    vo.vad_energy_threshold = energy_threshold;
    vo.vad_energy_mean_scale = energy_mean_scale;
    vo.vad_frames_context = frames_context;
    vo.vad_proportion_threshold = proportion_threshold;
    torch::Tensor f = torch::from_blob(const_cast<float *>(feats), {T, dim}, torch::kFloat).clone();
    torch::Tensor out;
    ex.ComputeVadEnergy(vo, f, out);
    out = out.contiguous();
    std::memcpy(voiced, out.data_ptr<float>(), sizeof(float) * (size_t)T);
    return 0;
  } catch (const std::exception &) {
    return -1;
  }
}
}
