// Drop-ins for two more reference tools, written against the same operator surface as HipOceanCurrents:
//   HipTectonics      for Tectonics (src/filter/tectonics/Tectonics.h:14-34), an AsyncSubFilter
//   HipGradientNoise  for GradientNoiseFilter (src/filter/GradientNoise.h:42-50), a one-step SubFilter
#pragma once
#include <string>
#include "Filter.h"
#include "../../include/demiurge_hip.h"

class HipTectonics : public AsyncSubFilter {
public:
    ~HipTectonics() override { stop(); }           // the worker is inside run(), reading the members below: joined before they go
    explicit HipTectonics(Project *p, int n_steps = 70) : n_steps_(n_steps) { (void)p; }   // 70: Tectonics.cpp:157
    const std::string &error() const { return error_; }

protected:
    void run() override;

private:
    int n_steps_;
    std::string error_;
};

class HipGradientNoise : public SubFilter {
public:
    // params = GradientNoiseMenu::params + the blend mode picked in the menu (GradientNoise.cpp:19,49)
    HipGradientNoise(const dmg_noise_params &params) : params_(params) {}
    std::pair<bool, float> step(Project *p) override;       // GradientNoiseFilter::step, :61-500
    const std::string &error() const { return error_; }

private:
    dmg_noise_params params_;
    std::string error_;
};
