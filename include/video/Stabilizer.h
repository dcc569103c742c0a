// vs::Stabilizer for MI355X - source-compatible replacement of the reference's
// include/video/Stabilizer.h (namespace vs, class Stabilizer, nested Parameters
// with the same field names, types and defaults: reference Stabilizer.h:25-198).
//
// Applications written against the reference (examples/vs.cpp:229,403,560,
// examples/file-capture.cpp:22-64, examples/vsg.cpp:1243-1289) compile
// unchanged: same constructor, stabilize()/flush()/clean(), value semantics
// (move-assignable; `stab = vs::Stabilizer(params)` works).  Everything behind
// the three calls runs on the GPU through the C ABI in include/vs_stab.h;
// OpenCV is used for cv::Mat I/O only.  The reference's ~20 private helper
// declarations (many without a definition) are not part of
// the interface and are not repeated here.
#ifndef VIDEO_STABILIZER_HPP
#define VIDEO_STABILIZER_HPP

#include <opencv2/opencv.hpp>

#include <string>

struct vs_stab;   // opaque C-ABI handle (include/vs_stab.h)

namespace vs {

    // reference Stabilizer.h:28-65
    struct Transform {
        float dx = 0.0f;
        float dy = 0.0f;
        float da = 0.0f;

        Transform() = default;
        Transform(float x, float y, float a) : dx(x), dy(y), da(a) {}
    };

    struct MotionSample {
        Transform transform;
        float magnitude = 0.0f;
        float confidence = 0.0f;
        int timestamp = 0;
    };

    enum class MotionType { NORMAL, INTENTIONAL_PAN, CAMERA_SHAKE, WALKING_VIBRATION, VEHICLE_VIBRATION };
    enum class MotionIntent { NORMAL, DELIBERATE_PAN, SHAKE_REMOVAL, FOLLOW_ACTION };
    enum class SceneType { NORMAL, SPORT, DRONE, HANDHELD, VEHICLE };

    class Stabilizer
    {
    public:
        // reference Stabilizer.h:76-175 - same names, types and defaults
        struct Parameters
        {
            bool useCuda = false;              ///< ignored: the GPU path is the only path of this build
            bool logging = false;

            int smoothingRadius = 30;
            int maxCorners = 200;
            double qualityLevel = 0.01;
            double minDistance = 30.0;
            int blockSize = 3;

            std::string borderType = "black";  ///< "black", "reflect", "reflect_101", "replicate", "wrap", "fade"
            int borderSize = 0;
            bool cropNZoom = false;

            std::string smoothingMethod = "box";  ///< "box", "gaussian", "kalman"
            double gaussianSigma = 2.0;
            bool motionPrediction = true;
            bool horizonLock = false;

            enum FeatureDetector { GFTT, ORB, FAST, BRISK };
            FeatureDetector featureDetector = GFTT;
            int orbFeatures = 500;
            int fastThreshold = 10;

            bool useROI = false;
            cv::Rect roi = cv::Rect();

            bool adaptiveSmoothing = false;
            int minSmoothingRadius = 5;
            int maxSmoothingRadius = 50;

            double outlierThreshold = 3.0;
            double intentionalMotionThreshold = 20.0;

            int stageOneRadius = 10;
            int stageTwoRadius = 25;
            bool useTemporalFiltering = false;
            int temporalWindowSize = 5;

            float fadeAlpha = 0.1f;
            int fadeDuration = 30;

            float motionThresholdLow = 5.0f;
            float motionThresholdHigh = 20.0f;
            float borderScaleFactor = 2.0f;

            bool rollCompensation = true;
            double rollCompensationFactor = 0.75;

            bool deepStabilization = false;
            std::string modelPath = "";

            enum JitterFrequency { LOW, MEDIUM, HIGH, ADAPTIVE };
            JitterFrequency jitterFrequency = ADAPTIVE;
            bool separateTranslationRotation = true;
            bool useImuData = false;

            bool enableVirtualCanvas = false;
            float canvasScaleFactor = 1.5f;
            int temporalBufferSize = 30;
            float canvasBlendWeight = 0.7f;
            bool adaptiveCanvasSize = true;
            float maxCanvasScale = 2.0f;
            float minCanvasScale = 1.2f;
            bool preserveEdgeQuality = true;
            int edgeBlendRadius = 20;

            bool droneHighFreqMode = false;
            float hfShakePx = 1.5f;
            int hfAnalysisMaxWidth = 960;
            float hfRotLPAlpha = 0.2f;
            bool enableConditionalCLAHE = true;

            float hfDeadZoneThreshold = 2.0f;
            int hfFreezeDuration = 10;
            float hfMotionAccumulatorDecay = 0.9f;

            // ---- additions of this implementation (behind the reference's fields, with defaults: code written against the
            // reference's struct compiles unchanged) -------------------------------------------------------------------------
            /// stabilize() returns the frame the PREVIOUS call made due: this call's upload, the device work and the download
            /// of that frame overlap (about twice the frame rate of the synchronous call); flush() hands out the held frame
            /// first.  One more call of latency, the same frames.  (The environment variable VS_STAB_HOST_PIPELINE=1 does the
            /// same for an application that cannot be rebuilt.)
            bool hostPipeline = false;
            /// The returned frames come from a small ring of page-locked cv::Mat buffers that are reused once the caller has let
            /// go of them (no allocation, no first-touch page faults, DMA straight into them).  false: a new cv::Mat per call.
            bool pinHostFrames = true;
            /// An INPUT buffer that keeps coming back (the Mat a capture loop reads into) is registered with the driver from
            /// its second appearance on, so that its upload is a DMA transfer of its own.  Off by default because the buffer is
            /// the application's: it must not be freed while it is registered - i.e. before clean(), the destruction of the
            /// Stabilizer, or 64 calls that did not use it.
            bool pinInputFrames = false;
        };

        explicit Stabilizer(const Parameters &params);
        ~Stabilizer();

        // The reference class is implicitly copyable; its call sites only ever
        // assign a freshly constructed temporary (vs.cpp:403,481).  Moves
        // transfer the GPU instance; a copy starts a new stream with the same
        // parameters (queued frames are not duplicated).
        Stabilizer(Stabilizer &&other) noexcept;
        Stabilizer &operator=(Stabilizer &&other) noexcept;
        Stabilizer(const Stabilizer &other);
        Stabilizer &operator=(const Stabilizer &other);

        /// BGR CV_8UC3 frame in; stabilized frame out, or an empty Mat while the
        /// first clamp(smoothingRadius,5,35)-1 frames are queued (reference Stabilizer.cpp:258-392).
        cv::Mat stabilize(const cv::Mat &frame);

        /// Next queued frame after the stream ended, or empty (Stabilizer.cpp:394-400).
        cv::Mat flush();

        /// Back to the first-frame state (Stabilizer.cpp:221-256).
        void clean();

    private:
        void logMessage(const std::string &msg, bool isError = false) const;
        void create();

        cv::Mat outputFrame(int rows, int cols);
        void noteInput(const cv::Mat &frame);
        void releaseHostPins();

        Parameters params_;
        vs_stab *impl_ = nullptr;
        int device_ = 0;
        int frameWidth_ = 0, frameHeight_ = 0;   // geometry of the stream (set by the first frame)
        // page-locked host memory (Parameters::pinHostFrames): ring of output frames, registered input buffers
        struct OutSlot { cv::Mat m; bool pinned = false; };
        struct InPin { const unsigned char *p = nullptr; size_t bytes = 0; int seen = 0; int idle = 0; bool pinned = false; };
        OutSlot outRing_[4];
        InPin inPins_[4];
    };

}

#endif // VIDEO_STABILIZER_HPP
