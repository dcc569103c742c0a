// Run-mode switches the reference's mains read from the "mode" section of config.yaml (examples/vs.cpp:59-68).
// Same field names, so code written against vs::Mode::Parameters compiles unchanged; defaults added (the reference
// leaves the fields uninitialised).
#pragma once

namespace vs {

class Mode {
public:
    struct Parameters {
        int width = 1920;
        int height = 1080;
        bool optimizeFps = false;
        bool useCuda = false;                // ignored by this build: the GPU path is the only one
        bool enhancerEnabled = false;
        bool rollCorrectionEnabled = false;
        bool stabilizationEnabled = false;
        bool trackerEnabled = false;         // object tracker: not part of this library
    };
};

}  // namespace vs
