// vs::Enhancer for MI355X - source-compatible with the reference's include/video/Enhancer.h:9-60:
// `vs::Enhancer::enhanceImage(frame, params)` and the `Parameters` field names are unchanged, so
// examples/vs.cpp:547-550 and its YAML reader (:71-104) compile as they are.  All stages run on the
// GPU through the C ABI (include/vs_stab.h, vs_enh_*); there is no CPU path in this build.
#ifndef VIDEO_ENHANCER_MI355X_HPP
#define VIDEO_ENHANCER_MI355X_HPP

#include <opencv2/core.hpp>

namespace vs {

class Enhancer {
public:
    struct Parameters {
        float brightness = 0.0f;            ///< added to every sample
        float contrast = 1.0f;              ///< multiplies every sample
        bool enableWhiteBalance = false;    ///< gray-world scaling of B, G, R ...
        float wbStrength = 1.0f;            ///< ... blended with identity by this factor
        bool enableVibrance = false;        ///< pushes HSV saturation towards 255 ...
        float vibranceStrength = 0.3f;      ///< ... by this fraction of the remaining headroom
        bool enableUnsharp = false;         ///< unsharp mask: img*(1+sharpness) - blur(img, blurSigma)*sharpness
        float sharpness = 0.0f;
        float blurSigma = 1.0f;
        bool enableClahe = false;           ///< CLAHE on L of Lab
        float claheClipLimit = 2.0f;
        int claheTileGridSize = 8;
        bool enableDenoise = false;         ///< cv::fastNlMeansDenoisingColored(h = hColor = denoiseStrength, 7, 21)
        float denoiseStrength = 10.0f;
        float gamma = 1.0f;                 ///< applied when |gamma - 1| > 1e-3
        /// The reference runs its stages in a different ORDER in its CUDA branch (Enhancer.cpp:183-233)
        /// than in its CPU branch (:142-181).  This build is always on the GPU; the flag only selects
        /// which of the two orders is reproduced.
        bool useCuda = false;
    };

    /// BGR8 in, enhanced BGR8 out (same size); an empty input gives an empty Mat (Enhancer.cpp:139-141).
    static cv::Mat enhanceImage(const cv::Mat& input, const Parameters& params);
};

}  // namespace vs

#endif
