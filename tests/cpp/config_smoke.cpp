// The config layer through its C++ face (include/video/Config.h): call forms of the reference's mains
// (examples/vs.cpp:50-131) against vs::ConfigFile, the bundled loader and the mtime watcher.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#include <sys/stat.h>
#include <utime.h>

#include "video/Config.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "config_smoke: %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string path = argv[1];

    // written like the mains
    vs::ConfigFile fs(path);
    CHECK(fs.isOpened());
    std::string videoSource;
    fs["video_source"] >> videoSource;
    CHECK(videoSource == "rtsp://192.168.144.119:554");
    vs::Mode::Parameters runParams;
    vs::ConfigNode modeNode = fs["mode"];
    CHECK(!modeNode.empty());
    modeNode["width"] >> runParams.width;
    modeNode["stabilizer_enabled"] >> runParams.stabilizationEnabled;
    modeNode["no_such_key"] >> runParams.height;                 // FileNode semantics: becomes 0
    CHECK(runParams.width == 1280 && runParams.stabilizationEnabled && runParams.height == 0);
    vs::Stabilizer::Parameters stabParams;
    vs::ConfigNode stabNode = fs["stabilizer"];
    stabNode["smoothing_radius"] >> stabParams.smoothingRadius;
    stabNode["border_type"] >> stabParams.borderType;
    stabNode["crop_n_zoom"] >> stabParams.cropNZoom;
    CHECK(stabParams.smoothingRadius == 15 && stabParams.borderType == "reflect_101" && !stabParams.cropNZoom);
    CHECK(fs["nothing"].empty() && fs["nothing"]["deeper"].empty());

    // strings longer than any fixed buffer
    {
        vs::ConfigFile big;
        const std::string longpath(9000, 'p');
        CHECK(big.parse("model_path: \"" + longpath + "\"\n"));
        std::string got;
        big["model_path"] >> got;
        CHECK(got == longpath);
    }

    // the bundled loader
    vs::AppConfig cfg;
    cfg.stabilizer.stageOneRadius = 77;                           // not in the file: must survive
    CHECK(vs::loadConfig(path, cfg));
    CHECK(cfg.mode.width == 1280 && cfg.mode.height == 720 && cfg.mode.enhancerEnabled && !cfg.mode.trackerEnabled);
    CHECK(cfg.mode.enhancerEnabled && cfg.mode.stabilizationEnabled && !cfg.mode.rollCorrectionEnabled && !cfg.mode.trackerEnabled);
    CHECK(cfg.enhancer.enableClahe && cfg.enhancer.claheTileGridSize == 4 && cfg.enhancer.blurSigma == 1.25f);
    CHECK(cfg.roll.houghThreshold == 80 && cfg.roll.scaleFactor == 0.5 && cfg.roll.maxAngleChangeDeg == 0.5);
    CHECK(cfg.stabilizer.maxCorners == 300 && cfg.stabilizer.smoothingMethod == "gaussian" && cfg.stabilizer.blockSize == 5);
    CHECK(cfg.stabilizer.droneHighFreqMode && cfg.stabilizer.hfAnalysisMaxWidth == 640);
    CHECK(cfg.stabilizer.stageOneRadius == 77);
    CHECK(cfg.stabilizer.jitterFrequency == vs::Stabilizer::Parameters::ADAPTIVE);
    vs::AppConfig strict;
    CHECK(vs::loadConfig(path, strict, /*keepMissing=*/false));
    CHECK(strict.stabilizer.stageOneRadius == 0 && strict.stabilizer.maxCorners == 300);

    // a broken edit leaves the configuration alone
    const std::string bad = path + ".bad";
    { std::ofstream f(bad); f << "stabilizer:\n\tsmoothing_radius: 3\n"; }
    std::string err;
    vs::AppConfig keep = cfg;
    CHECK(!vs::loadConfig(bad, keep, true, &err) && err.find("line 2") != std::string::npos);
    CHECK(keep.stabilizer.smoothingRadius == 15);

    // hot reload
    vs::ConfigWatcher watch(path);
    CHECK(!watch.changed());
    struct stat st;
    CHECK(stat(path.c_str(), &st) == 0);
    struct utimbuf t = {st.st_atime, st.st_mtime + 3};
    CHECK(utime(path.c_str(), &t) == 0);
    CHECK(watch.changed() && !watch.changed());
    printf("config ok\n");
    return 0;
}
