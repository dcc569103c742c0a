// Config layer of libvideo-stab: the YAML the reference's example mains read with
// cv::FileStorage (examples/config.yaml; call sites examples/vs.cpp:50-168, examples/vsg.cpp:1007-1112) and the
// key -> parameter mapping every one of them repeats, as one reusable host-side helper.  No device work.
//
// cv::FileStorage is OpenCV (third party, not vendored by the reference and not present in this image).  What is
// restated here is the published behaviour of its YAML reader and of `FileNode >> value`
// (opencv2/core/persistence.hpp, modules/core/src/persistence_yml.cpp, OpenCV 4.x), for the subset of YAML those
// config files use: nested block maps, scalars, quoted strings, comments, flow sequences of scalars.
//   scalars : true/True/TRUE -> int 1, false/False/FALSE -> int 0; a token that starts like a number is an int
//             unless a '.' or 'e' follows its leading digits (then a real); everything else is a string
//   >> int    : int as is, real rounded half to even (cvRound), missing 0, anything else INT_MAX
//   >> double : int / real as is, missing 0, anything else DBL_MAX (FLT_MAX for float)
//   >> string : string as is, missing or anything else ""
//   >> bool   : (>> int) != 0
#include <sys/stat.h>

#include <cerrno>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "vs_common.h"

namespace {

struct Node {
    int kind = VS_CFG_NONE;
    long long ival = 0;
    double dval = 0;
    std::string sval;
    std::vector<std::pair<std::string, Node>> map;
    std::vector<Node> seq;
    const Node* find(const std::string& key) const {
        for (const auto& kv : map) if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

struct Line {
    int indent = 0, number = 0;
    std::string text;       // without indentation, comment and trailing blanks
};

bool fail(int line, const char* what) {
    char buf[160];
    snprintf(buf, sizeof buf, "config: line %d: %s", line, what);
    vsd::set_last_error(buf);
    return false;
}

// the text up to a comment that is outside quotes ('#' at the start or after a blank)
std::string strip_comment(const std::string& s) {
    char quote = 0;
    size_t end = s.size();
    for (size_t i = 0; i < s.size(); i++) {
        const char c = s[i];
        if (quote) {
            if (c == '\\' && quote == '"') i++;
            else if (c == quote) quote = 0;
        } else if (c == '"' || c == '\'') {
            quote = c;
        } else if (c == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) {
            end = i;
            break;
        }
    }
    while (end > 0 && (s[end - 1] == ' ' || s[end - 1] == '\t' || s[end - 1] == '\r')) end--;
    return s.substr(0, end);
}

bool unquote(const std::string& t, int line, std::string& out) {
    const char q = t[0];
    if (t.size() < 2 || t.back() != q) return fail(line, "unterminated quoted string");
    out.clear();
    for (size_t i = 1; i + 1 < t.size(); i++) {
        char c = t[i];
        if (q == '"' && c == '\\' && i + 2 < t.size()) {
            c = t[++i];
            if (c == 'n') c = '\n';
            else if (c == 't') c = '\t';
            else if (c == 'r') c = '\r';
            else if (c == '0') c = '\0';
        } else if (q == '\'' && c == '\'' && i + 2 < t.size() && t[i + 1] == '\'') {
            i++;
        }
        out.push_back(c);
    }
    return true;
}

bool parse_scalar(const std::string& t, int line, Node& n) {
    n = Node();
    if (t.empty() || t == "~" || t == "null" || t == "NULL" || t == "Null") return true;
    if (t[0] == '"' || t[0] == '\'') {
        n.kind = VS_CFG_STRING;
        return unquote(t, line, n.sval);
    }
    if (t == "true" || t == "True" || t == "TRUE") { n.kind = VS_CFG_INT; n.ival = 1; return true; }
    if (t == "false" || t == "False" || t == "FALSE") { n.kind = VS_CFG_INT; n.ival = 0; return true; }
    const char c = t[0], d = t.size() > 1 ? t[1] : '\0';
    const bool numeric = isdigit((unsigned char)c) || ((c == '-' || c == '+') && (isdigit((unsigned char)d) || d == '.')) ||
                         (c == '.' && isalnum((unsigned char)d));
    if (numeric) {
        if (t == ".inf" || t == ".Inf" || t == ".INF" || t == "+.inf" || t == "+.Inf" || t == "+.INF") { n.kind = VS_CFG_REAL; n.dval = INFINITY; return true; }
        if (t == "-.inf" || t == "-.Inf" || t == "-.INF") { n.kind = VS_CFG_REAL; n.dval = -INFINITY; return true; }
        if (t == ".nan" || t == ".NaN" || t == ".NAN") { n.kind = VS_CFG_REAL; n.dval = NAN; return true; }
        size_t i = (c == '-' || c == '+') ? 1 : 0;
        while (i < t.size() && isdigit((unsigned char)t[i])) i++;
        char* end = nullptr;
        errno = 0;
        if (i < t.size() && (t[i] == '.' || t[i] == 'e' || t[i] == 'E')) {
            const double v = strtod(t.c_str(), &end);
            if (end && *end == '\0') { n.kind = VS_CFG_REAL; n.dval = v; return true; }
        } else {
            const long long v = strtoll(t.c_str(), &end, 0);
            if (end && *end == '\0' && errno == 0) { n.kind = VS_CFG_INT; n.ival = v; return true; }
        }
        // something like 192.168.1.1 or 10px: not a number after all
    }
    n.kind = VS_CFG_STRING;
    n.sval = t;
    return true;
}

bool parse_flow_seq(const std::string& t, int line, Node& n) {
    if (t.back() != ']') return fail(line, "flow sequence must close on its own line");
    n = Node();
    n.kind = VS_CFG_SEQ;
    std::string item;
    char quote = 0;
    auto flush = [&](bool last) {
        size_t a = 0, b = item.size();
        while (a < b && (item[a] == ' ' || item[a] == '\t')) a++;
        while (b > a && (item[b - 1] == ' ' || item[b - 1] == '\t')) b--;
        if (a == b && last && n.seq.empty()) return true;           // []
        Node e;
        if (!parse_scalar(item.substr(a, b - a), line, e)) return false;
        n.seq.push_back(e);
        item.clear();
        return true;
    };
    for (size_t i = 1; i + 1 < t.size(); i++) {
        const char c = t[i];
        if (quote) { if (c == quote) quote = 0; item.push_back(c); }
        else if (c == '"' || c == '\'') { quote = c; item.push_back(c); }
        else if (c == '[' || c == '{') return fail(line, "nested flow collections are not supported");
        else if (c == ',') { if (!flush(false)) return false; }
        else item.push_back(c);
    }
    return flush(true);
}

bool parse_value(const std::string& t, int line, Node& n) {
    if (!t.empty() && t[0] == '[') return parse_flow_seq(t, line, n);
    if (!t.empty() && t[0] == '{') {
        if (t == "{}") { n = Node(); n.kind = VS_CFG_MAP; return true; }
        return fail(line, "flow maps are not supported");
    }
    return parse_scalar(t, line, n);
}

// "key: value" / "key:" -> key, rest.  The colon must be followed by a blank or end the line.
bool split_key(const std::string& t, int line, std::string& key, std::string& rest, bool& is_pair) {
    is_pair = false;
    size_t i = 0;
    if (t[0] == '"' || t[0] == '\'') {
        const char q = t[0];
        i = 1;
        while (i < t.size() && t[i] != q) i++;
        if (i >= t.size()) return fail(line, "unterminated quoted key");
        i++;
    } else {
        while (i < t.size() && !(t[i] == ':' && (i + 1 == t.size() || t[i + 1] == ' ' || t[i + 1] == '\t'))) i++;
    }
    if (i >= t.size() || t[i] != ':') return true;
    std::string k = t.substr(0, i);
    if (k[0] == '"' || k[0] == '\'') { if (!unquote(k, line, key)) return false; }
    else key = k;
    size_t j = i + 1;
    while (j < t.size() && (t[j] == ' ' || t[j] == '\t')) j++;
    rest = t.substr(j);
    is_pair = true;
    return true;
}

bool parse_block(const std::vector<Line>& L, size_t& at, int indent, Node& out);

bool parse_map(const std::vector<Line>& L, size_t& at, int indent, Node& out) {
    out = Node();
    out.kind = VS_CFG_MAP;
    while (at < L.size() && L[at].indent == indent) {
        const Line& ln = L[at];
        std::string key, rest;
        bool is_pair;
        if (ln.text[0] == '-' && (ln.text.size() == 1 || ln.text[1] == ' ')) return fail(ln.number, "sequence item inside a map");
        if (!split_key(ln.text, ln.number, key, rest, is_pair)) return false;
        if (!is_pair) return fail(ln.number, "expected 'key: value'");
        if (out.find(key)) return fail(ln.number, "duplicate key");
        Node v;
        at++;
        if (rest.empty()) {
            if (at < L.size() && L[at].indent > indent) {
                if (!parse_block(L, at, L[at].indent, v)) return false;
            } else if (at < L.size() && L[at].indent == indent && L[at].text[0] == '-' &&
                       (L[at].text.size() == 1 || L[at].text[1] == ' ')) {
                if (!parse_block(L, at, indent, v)) return false;       // a sequence may sit at its key's indentation
            }
        } else if (!parse_value(rest, ln.number, v)) {
            return false;
        }
        out.map.emplace_back(key, std::move(v));
    }
    if (at < L.size() && L[at].indent > indent) return fail(L[at].number, "unexpected indentation");
    return true;
}

bool parse_seq(const std::vector<Line>& L, size_t& at, int indent, Node& out) {
    out = Node();
    out.kind = VS_CFG_SEQ;
    while (at < L.size() && L[at].indent == indent && L[at].text[0] == '-' && (L[at].text.size() == 1 || L[at].text[1] == ' ')) {
        const Line& ln = L[at];
        size_t j = 1;
        while (j < ln.text.size() && ln.text[j] == ' ') j++;
        const std::string rest = ln.text.substr(j);
        Node v;
        at++;
        if (rest.empty()) {
            if (at < L.size() && L[at].indent > indent && !parse_block(L, at, L[at].indent, v)) return false;
        } else {
            std::string key, r2;
            bool is_pair;
            if (rest[0] != '[' && rest[0] != '{') {
                if (!split_key(rest, ln.number, key, r2, is_pair)) return false;
                if (is_pair) return fail(ln.number, "maps inside sequences are not supported");
            }
            if (!parse_value(rest, ln.number, v)) return false;
        }
        out.seq.push_back(std::move(v));
    }
    return true;
}

bool parse_block(const std::vector<Line>& L, size_t& at, int indent, Node& out) {
    const std::string& t = L[at].text;
    if (t[0] == '-' && (t.size() == 1 || t[1] == ' ')) return parse_seq(L, at, indent, out);
    return parse_map(L, at, indent, out);
}

bool parse_text(const char* text, size_t len, Node& root) {
    std::vector<Line> L;
    int number = 0;
    size_t i = 0;
    while (i < len) {
        size_t e = i;
        while (e < len && text[e] != '\n') e++;
        std::string raw(text + i, e - i);
        i = e + 1;
        number++;
        if (number == 1 && raw.size() >= 3 && (unsigned char)raw[0] == 0xEF && (unsigned char)raw[1] == 0xBB && (unsigned char)raw[2] == 0xBF)
            raw.erase(0, 3);
        int indent = 0;
        while ((size_t)indent < raw.size() && raw[indent] == ' ') indent++;
        if ((size_t)indent < raw.size() && raw[indent] == '\t') return fail(number, "tab in indentation");
        const std::string body = strip_comment(raw.substr(indent));
        if (body.empty()) continue;
        if (indent == 0 && (body[0] == '%' || body == "---" || body == "...")) continue;
        Line ln;
        ln.indent = indent; ln.number = number; ln.text = body;
        L.push_back(ln);
    }
    root = Node();
    root.kind = VS_CFG_MAP;
    if (L.empty()) return true;
    size_t at = 0;
    if (!parse_block(L, at, L[0].indent, root)) return false;
    if (at < L.size()) return fail(L[at].number, "unexpected indentation");
    if (root.kind != VS_CFG_MAP) return fail(L[0].number, "top level must be a map");
    return true;
}

const Node* lookup(const Node* n, const char* path) {
    if (!path) return nullptr;
    const char* p = path;
    while (n && *p) {
        const char* dot = strchr(p, '.');
        const std::string key = dot ? std::string(p, dot - p) : std::string(p);
        if (n->kind != VS_CFG_MAP) return nullptr;
        n = n->find(key);
        if (!dot) break;
        p = dot + 1;
    }
    return n;
}

int to_int(const Node* n) {
    if (!n || n->kind == VS_CFG_NONE) return 0;
    if (n->kind == VS_CFG_INT) return (int)n->ival;
    if (n->kind == VS_CFG_REAL) return (int)lrint(n->dval);       // cvRound: to nearest, ties to even
    return INT_MAX;
}

double to_double(const Node* n) {
    if (!n || n->kind == VS_CFG_NONE) return 0.0;
    if (n->kind == VS_CFG_INT) return (double)n->ival;
    if (n->kind == VS_CFG_REAL) return n->dval;
    return DBL_MAX;
}

float to_float(const Node* n) {
    if (!n || n->kind == VS_CFG_NONE) return 0.f;
    if (n->kind == VS_CFG_INT) return (float)n->ival;
    if (n->kind == VS_CFG_REAL) return (float)n->dval;
    return FLT_MAX;
}

}  // namespace

struct vs_config {
    Node root;
};

namespace {

// `section[key] >> field`, or nothing when the key is absent and the caller asked to keep what it has
struct Reader {
    const Node* sec;
    bool zero_missing;
    const Node* get(const char* key, bool& apply) const {
        const Node* n = sec->find(key);
        apply = n != nullptr || zero_missing;
        return n;
    }
    void i32(const char* key, int32_t& f) const { bool a; const Node* n = get(key, a); if (a) f = to_int(n); }
    void flag(const char* key, int32_t& f) const { bool a; const Node* n = get(key, a); if (a) f = to_int(n) != 0; }
    void f64(const char* key, double& f) const { bool a; const Node* n = get(key, a); if (a) f = to_double(n); }
    void f32(const char* key, float& f) const { bool a; const Node* n = get(key, a); if (a) f = to_float(n); }
    bool str(const char* key, std::string& s) const {
        bool a; const Node* n = get(key, a);
        if (a) s = (n && n->kind == VS_CFG_STRING) ? n->sval : std::string();
        return a;
    }
};

const Node* section_of(const vs_config* c, const char* section) {
    if (!c) return nullptr;
    const Node* n = section && *section ? lookup(&c->root, section) : &c->root;
    return (n && n->kind == VS_CFG_MAP) ? n : nullptr;
}

}  // namespace

extern "C" {

int vs_config_parse(const char* text, size_t len, vs_config** out) {
    if (!out || (!text && len)) return VS_ERR_INVALID_ARG;
    *out = nullptr;
    vs_config* c = new (std::nothrow) vs_config();
    if (!c) return VS_ERR_HIP;
    if (!parse_text(text ? text : "", len, c->root)) { delete c; return VS_ERR_INVALID_ARG; }
    *out = c;
    return VS_OK;
}

int vs_config_open(const char* path, vs_config** out) {
    if (!path || !out) return VS_ERR_INVALID_ARG;
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) { vsd::set_last_error(std::string("config: cannot open ") + path); return VS_ERR_INVALID_ARG; }
    std::string text;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    fclose(f);
    return vs_config_parse(text.data(), text.size(), out);
}

void vs_config_close(vs_config* c) { delete c; }

int vs_config_kind(const vs_config* c, const char* key_path) {
    if (!c) return VS_CFG_NONE;
    const Node* n = lookup(&c->root, key_path);
    return n ? n->kind : VS_CFG_NONE;
}

int vs_config_size(const vs_config* c, const char* key_path) {
    if (!c) return 0;
    const Node* n = lookup(&c->root, key_path);
    if (!n) return 0;
    if (n->kind == VS_CFG_MAP) return (int)n->map.size();
    if (n->kind == VS_CFG_SEQ) return (int)n->seq.size();
    return n->kind == VS_CFG_NONE ? 0 : 1;
}

int vs_config_get_int(const vs_config* c, const char* key_path, int32_t* v) {
    if (!c || !v) return VS_ERR_INVALID_ARG;
    *v = to_int(lookup(&c->root, key_path));
    return VS_OK;
}

int vs_config_get_double(const vs_config* c, const char* key_path, double* v) {
    if (!c || !v) return VS_ERR_INVALID_ARG;
    *v = to_double(lookup(&c->root, key_path));
    return VS_OK;
}

int vs_config_get_float(const vs_config* c, const char* key_path, float* v) {
    if (!c || !v) return VS_ERR_INVALID_ARG;
    *v = to_float(lookup(&c->root, key_path));
    return VS_OK;
}

int vs_config_get_string(const vs_config* c, const char* key_path, char* buf, size_t cap) {
    if (!c || !buf || cap == 0) return VS_ERR_INVALID_ARG;
    const Node* n = lookup(&c->root, key_path);
    const std::string s = (n && n->kind == VS_CFG_STRING) ? n->sval : std::string();
    if (s.size() + 1 > cap) { vsd::set_last_error("config: string does not fit the buffer"); return VS_ERR_INVALID_ARG; }
    memcpy(buf, s.c_str(), s.size() + 1);
    return VS_OK;
}

int vs_config_seq_get_double(const vs_config* c, const char* key_path, int index, double* v) {
    if (!c || !v) return VS_ERR_INVALID_ARG;
    const Node* n = lookup(&c->root, key_path);
    if (!n || n->kind != VS_CFG_SEQ || index < 0 || index >= (int)n->seq.size()) return VS_ERR_INVALID_ARG;
    *v = to_double(&n->seq[index]);
    return VS_OK;
}

int vs_config_read_stab(const vs_config* c, const char* section, int zero_missing, vs_params_c* p, int* present) {
    if (!c || !p || p->struct_size != (int32_t)sizeof(vs_params_c)) return VS_ERR_INVALID_ARG;
    const Node* sec = section_of(c, section);
    if (present) *present = sec != nullptr;
    if (!sec) return VS_OK;                                  // `if (!stabNode.empty())`: an absent section changes nothing
    const Reader r{sec, zero_missing != 0};
    r.i32("smoothing_radius", p->smoothing_radius);
    std::string s;
    if (r.str("border_type", s)) {
        p->border_type = s == "reflect" ? VS_BORDER_REFLECT : s == "reflect_101" ? VS_BORDER_REFLECT_101
                         : s == "replicate" ? VS_BORDER_REPLICATE : s == "wrap" ? VS_BORDER_WRAP
                         : s == "fade" ? VS_BORDER_FADE : VS_BORDER_BLACK;
    }
    r.i32("border_size", p->border_size);
    r.flag("crop_n_zoom", p->crop_n_zoom);
    r.flag("logging", p->logging);
    if (r.str("smoothing_method", s))
        p->smoothing_method = s == "gaussian" ? VS_SMOOTH_GAUSSIAN : s == "kalman" ? VS_SMOOTH_KALMAN : VS_SMOOTH_BOX;
    r.f64("gaussian_sigma", p->gaussian_sigma);
    r.flag("adaptive_smoothing", p->adaptive_smoothing);
    r.i32("min_smoothing_radius", p->min_smoothing_radius);
    r.i32("max_smoothing_radius", p->max_smoothing_radius);
    r.i32("max_corners", p->max_corners);
    r.f64("quality_level", p->quality_level);
    r.f64("min_distance", p->min_distance);
    r.i32("block_size", p->block_size);
    r.flag("horizon_lock", p->horizon_lock);
    r.i32("fadeDuration", p->fade_duration);
    r.f32("fadeAlpha", p->fade_alpha);
    r.flag("enable_virtual_canvas", p->enable_virtual_canvas);     // examples/vsg.cpp:1086-1094
    r.f32("canvas_scale_factor", p->canvas_scale_factor);
    r.i32("temporal_buffer_size", p->temporal_buffer_size);
    r.f32("canvas_blend_weight", p->canvas_blend_weight);
    r.flag("adaptive_canvas_size", p->adaptive_canvas_size);
    r.f32("max_canvas_scale", p->max_canvas_scale);
    r.f32("min_canvas_scale", p->min_canvas_scale);
    r.i32("edge_blend_radius", p->edge_blend_radius);
    r.flag("drone_high_freq_mode", p->drone_high_freq_mode);
    r.f32("hf_shake_px", p->hf_shake_px);
    r.i32("hf_analysis_max_width", p->hf_analysis_max_width);
    r.f32("hf_rot_lp_alpha", p->hf_rot_lp_alpha);
    r.flag("enable_conditional_clahe", p->enable_conditional_clahe);
    r.f32("hf_dead_zone_threshold", p->hf_dead_zone_threshold);
    r.i32("hf_freeze_duration", p->hf_freeze_duration);
    r.f32("hf_motion_accumulator_decay", p->hf_motion_accumulator_decay);
    // Stabilizer.cpp:67-71: crop-and-zoom only works on a black border
    if (p->crop_n_zoom && p->border_type != VS_BORDER_BLACK) p->border_type = VS_BORDER_BLACK;
    return VS_OK;
}

int vs_config_read_roll(const vs_config* c, const char* section, int zero_missing, vs_roll_params_c* p, int* present) {
    if (!c || !p || p->struct_size != (int32_t)sizeof(vs_roll_params_c)) return VS_ERR_INVALID_ARG;
    const Node* sec = section_of(c, section);
    if (present) *present = sec != nullptr;
    if (!sec) return VS_OK;
    const Reader r{sec, zero_missing != 0};
    r.f64("scale_factor", p->scale_factor);
    r.f64("canny_threshold_low", p->canny_threshold_low);
    r.f64("canny_threshold_high", p->canny_threshold_high);
    r.i32("canny_aperture", p->canny_aperture);
    r.f32("hough_rho", p->hough_rho);
    r.f32("hough_theta", p->hough_theta);
    r.i32("hough_threshold", p->hough_threshold);
    r.f64("angle_smoothing_alpha", p->angle_smoothing_alpha);
    r.f64("angle_decay", p->angle_decay);
    r.f64("angle_filter_min", p->angle_filter_min);
    r.f64("angle_filter_max", p->angle_filter_max);
    return VS_OK;
}

int vs_config_read_enh(const vs_config* c, const char* section, int zero_missing, vs_enh_params_c* p, int* present) {
    if (!c || !p || p->struct_size != (int32_t)sizeof(vs_enh_params_c)) return VS_ERR_INVALID_ARG;
    const Node* sec = section_of(c, section);
    if (present) *present = sec != nullptr;
    if (!sec) return VS_OK;
    const Reader r{sec, zero_missing != 0};
    r.f32("brightness", p->brightness);
    r.f32("contrast", p->contrast);
    r.flag("enable_white_balance", p->enable_white_balance);
    r.f32("wb_strength", p->wb_strength);
    r.flag("enable_vibrance", p->enable_vibrance);
    r.f32("vibrance_strength", p->vibrance_strength);
    r.flag("enable_unsharp", p->enable_unsharp);
    r.f32("sharpness", p->sharpness);
    r.f32("blur_sigma", p->blur_sigma);
    r.flag("enable_denoise", p->enable_denoise);
    r.f32("denoise_strength", p->denoise_strength);
    r.f32("gamma", p->gamma);
    r.flag("enable_clahe", p->enable_clahe);
    r.f32("clahe_clip_limit", p->clahe_clip_limit);
    r.i32("clahe_tile_grid_size", p->clahe_tile_grid_size);
    r.flag("use_cuda", p->use_cuda);
    return VS_OK;
}

// st_mtime of the file (the hot-reload test of the example mains, vs.cpp:199-200,381-394); VS_ERR_INVALID_ARG when it
// cannot be stat'ed.
int vs_config_mtime(const char* path, int64_t* mtime) {
    if (!path || !mtime) return VS_ERR_INVALID_ARG;
    struct stat st;
    if (stat(path, &st) != 0) { vsd::set_last_error(std::string("config: cannot stat ") + path); return VS_ERR_INVALID_ARG; }
    *mtime = (int64_t)st.st_mtime;
    return VS_OK;
}

}  // extern "C"
