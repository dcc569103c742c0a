// Host-side robustness check of the config reader: csrc/config.cpp built for the CPU with ASan + UBSan and fed
// mutated documents.  g++ -fsanitize=address,undefined ... (see scratch/README.md).  Not part of the product.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>

#include "vs_stab.h"

namespace vsd {
static std::string g_err;
void set_last_error(const std::string& m) { g_err = m; }
}

static const char* SEED_DOC =
    "%YAML:1.0\nvideo_source: \"rtsp://h:554\"   # c\nmode:\n  width:  1280\n  use_cuda: true\nstabilizer:\n"
    "  smoothing_radius: 15\n  border_type: \"reflect_101\"\n  gaussian_sigma: 15.0\n  roi: [1, 2.5, 'x', \"y\"]\n"
    "  'quoted key': 'it''s'\nlist:\n  - 1\n  - -2.5e3\n  - .inf\nempty: []\nm: {}\nk:\n- 3\n- 4\n";

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200000;
    std::mt19937 rng(12345);
    const std::string seed = SEED_DOC;
    const char alphabet[] = " \t\n:#-[]{},'\"\\.%~0123456789eExXabz+";
    long ok = 0, bad = 0;
    for (int it = 0; it < rounds; it++) {
        std::string doc = seed;
        const int edits = 1 + rng() % 6;
        for (int e = 0; e < edits; e++) {
            const size_t at = doc.empty() ? 0 : rng() % doc.size();
            switch (rng() % 4) {
                case 0: if (!doc.empty()) doc.erase(at, 1 + rng() % 4); break;
                case 1: doc.insert(at, 1, alphabet[rng() % (sizeof alphabet - 1)]); break;
                case 2: if (!doc.empty()) doc[at] = alphabet[rng() % (sizeof alphabet - 1)]; break;
                default: if (!doc.empty()) doc.insert(at, doc.substr(rng() % doc.size(), rng() % 12)); break;
            }
        }
        if (rng() % 50 == 0) doc.resize(rng() % (doc.size() + 1));
        vs_config* c = nullptr;
        if (vs_config_parse(doc.data(), doc.size(), &c) == VS_OK) {
            ok++;
            vs_params_c p;
            vs_params_default(&p);
            int present = 0;
            vs_config_read_stab(c, "stabilizer", rng() & 1, &p, &present);
            int32_t i; double d; float f; char buf[64];
            vs_config_get_int(c, "stabilizer.roi", &i);
            vs_config_get_double(c, "mode.width", &d);
            vs_config_get_float(c, "list", &f);
            vs_config_get_string(c, "video_source", buf, sizeof buf);
            for (int k = -1; k < 6; k++) vs_config_seq_get_double(c, "stabilizer.roi", k, &d);
            vs_config_size(c, "stabilizer"); vs_config_kind(c, "a.b.c.d"); vs_config_kind(c, ""); vs_config_kind(c, "..");
            vs_config_close(c);
        } else {
            bad++;
            if (vsd::g_err.find("config: line") != 0) { fprintf(stderr, "unexpected error text: %s\n", vsd::g_err.c_str()); return 1; }
        }
    }
    printf("%d documents: %ld parsed, %ld rejected with a line number\n", rounds, ok, bad);
    return 0;
}
