// Drop-ins for the remaining simulation tools of the Filter menu, written against the same operator surface as
// HipOceanCurrents (Filter.h here = src/filter/Filter.h:58-115 without GL):
//   HipErosion        cpufilter             (src/filter/cpufilter.h:25-43)         AsyncSubFilter
//   HipTemperature    Temperature           (src/filter/Temperature.h:22-27)       AsyncSubFilter
//   HipFlow           FlowFilter            (src/filter/FlowFilter.h:25-29)        AsyncSubFilter
//   HipBlur           Blur                  (src/filter/BlurMenu.h:13-24)          SubFilter
//   HipThermalErosion ThermalErosion        (src/filter/ThermalErosion.h:12-15)    SubFilter
//   HipMorphological  Morphological / MorphologicalGradient (src/filter/Morphological.h:24-38)   SubFilter
// Each keeps the reference class's constructor arguments and its step()/run() protocol; the target layer is the
// terrain (what every makeFilter of these tools passes, e.g. cpufilter.cpp:30-31).
#pragma once
#include <string>
#include "Filter.h"
#include "../../include/demiurge_hip.h"

class HipErosion : public AsyncSubFilter {
public:
    ~HipErosion() override { stop(); }           // the worker is inside run(), reading the members below: joined before they go
    // cpufilter(Project*, exponent, slope_exponent, factor, dolakes); N = 50 and 3*N rounds are the reference's (:42,:93)
    HipErosion(Project *p, float exponent, float slope_exponent, float factor, int dolakes, int N = 50, int rounds = 150)
        : exponent_(exponent), slope_exponent_(slope_exponent), factor_(factor), dolakes_(dolakes), N_(N), rounds_(rounds) { (void)p; }
    const std::string &error() const { return error_; }
protected:
    void run() override;
private:
    float exponent_, slope_exponent_, factor_;
    int dolakes_, N_, rounds_;
    std::string error_;
};

class HipTemperature : public AsyncSubFilter {
public:
    ~HipTemperature() override { stop(); }           // the worker is inside run(), reading the members below: joined before they go
    // Temperature(Project*) (Temperature.cpp:26-44): the heights move to scratch1 (terrain.swap(scratch1)) and the terrain
    // layer becomes the temperature field, 50 everywhere.  The reference then runs 500000 frames of 10 steps (:48,:140)
    // -- `frames` bounds it here.  (Its closing terrain.swap(scratch2), :158-161, is never reached in practice and is
    // not mirrored: the terrain layer keeps the last field.)
    explicit HipTemperature(Project *p, int frames = 500000, int steps_per_frame = 10) : frames_(frames), per_frame_(steps_per_frame)
    {
        std::swap(p->terrain, p->scratch1);
        p->terrain.assign((size_t)p->getWidth() * p->getHeight(), 50.0f);
        dmg_climate_params_default(&params_);
    }
    // the shader's literals (Temperature.cpp:56-134) as the menu's fields: set before the first step()
    dmg_climate_params &params() { return params_; }
    const std::string &error() const { return error_; }
protected:
    void run() override;
private:
    int frames_, per_frame_;
    dmg_climate_params params_;
    std::string error_;
};

class HipFlow : public AsyncSubFilter {
public:
    ~HipFlow() override { stop(); }           // the worker is inside run(), reading the members below: joined before they go
    HipFlow(float preblur, float exponent, bool lakeflag = false) : preblur_(preblur), exponent_(exponent), lakeflag_(lakeflag) {}   // FlowFilter.cpp:29-33
    const std::string &error() const { return error_; }
protected:
    void run() override;
private:
    float preblur_, exponent_;
    bool lakeflag_;
    std::string error_;
};

// one-call tools: the reference spreads their passes over several frames only to keep the UI alive; the result of the
// last frame is what counts, and one device call produces it
class HipOneShot : public SubFilter {
public:
    std::pair<bool, float> step(Project *p) override;
    const std::string &error() const { return error_; }
protected:
    virtual int apply(dmg_ctx *ctx, const dmg_grid *g, float *terrain) = 0;
    std::string error_;
};
class HipBlur : public HipOneShot {
public:
    HipBlur(Project *p, float radius) : radius_(radius) { (void)p; }                      // Blur(p, radius, target), BlurMenu.cpp:24
protected:
    int apply(dmg_ctx *ctx, const dmg_grid *g, float *t) override { return dmg_blur(ctx, g, t, radius_); }
private:
    float radius_;
};
class HipThermalErosion : public HipOneShot {
protected:
    int apply(dmg_ctx *ctx, const dmg_grid *g, float *t) override { return dmg_thermal_erosion(ctx, g, t); }   // ThermalErosion.cpp:25-142
};
class HipMorphological : public HipOneShot {
public:
    HipMorphological(Project *p, float radius, const std::string &operation) : radius_(radius), op_(operation == "max" ? 1 : 0) { (void)p; }   // Morphological.cpp:28
protected:
    int apply(dmg_ctx *ctx, const dmg_grid *g, float *t) override { return dmg_morphological(ctx, g, t, radius_, op_); }
private:
    float radius_;
    int op_;
};
class HipMorphologicalGradient : public HipOneShot {
public:
    HipMorphologicalGradient(Project *p, float radius) : radius_(radius) { (void)p; }     // Morphological.cpp:87
protected:
    int apply(dmg_ctx *ctx, const dmg_grid *g, float *t) override { return dmg_morphological_gradient(ctx, g, t, radius_); }
private:
    float radius_;
};
