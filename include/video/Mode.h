// Run-mode switches of the example applications: which stages of the chain
// enhance -> roll correction -> stabilize -> track are on, and the size the
// pipeline runs at.  Field names are those the reference's mains assign from
// the "mode" section of config.yaml (examples/vs.cpp:59-68), so code written
// against vs::Mode::Parameters compiles unchanged; the struct carries defaults
// here (everything off, 1080p), which the reference leaves uninitialised.
#pragma once

#include <string>

namespace vs {

class Mode {
public:
    struct Parameters {
        int width = 1920;                    ///< frames are resized to width x height on ingest
        int height = 1080;
        bool optimizeFps = false;            ///< drop work to hold the frame rate
        bool useCuda = false;                ///< ignored by this build: the GPU path is the only one
        bool enhancerEnabled = false;        ///< vs::Enhancer::enhanceImage
        bool rollCorrectionEnabled = false;  ///< vs::RollCorrection::autoCorrectRoll
        bool stabilizationEnabled = false;   ///< vs::Stabilizer::stabilize
        bool trackerEnabled = false;         ///< object tracker (not part of this library)

        /// Stages of this library that are switched on.
        int enabledStages() const { return (int)enhancerEnabled + (int)rollCorrectionEnabled + (int)stabilizationEnabled; }

        /// One line for a log, e.g. "1920x1080 enhance+stabilize".
        std::string describe() const {
            std::string chain;
            auto add = [&chain](bool on, const char* name) {
                if (!on) return;
                if (!chain.empty()) chain += '+';
                chain += name;
            };
            add(enhancerEnabled, "enhance");
            add(rollCorrectionEnabled, "roll");
            add(stabilizationEnabled, "stabilize");
            add(trackerEnabled, "track");
            return std::to_string(width) + "x" + std::to_string(height) + " " + (chain.empty() ? "passthrough" : chain);
        }

        bool operator==(const Parameters& o) const {
            return width == o.width && height == o.height && optimizeFps == o.optimizeFps && useCuda == o.useCuda &&
                   enhancerEnabled == o.enhancerEnabled && rollCorrectionEnabled == o.rollCorrectionEnabled &&
                   stabilizationEnabled == o.stabilizationEnabled && trackerEnabled == o.trackerEnabled;
        }
        bool operator!=(const Parameters& o) const { return !(*this == o); }
    };
};

}  // namespace vs
