// host_util.cpp - error plumbing, minimal JSON reader, safetensors mmap reader.
// Replaces (host side) MLX.loadArrays + JSONDecoder use in SNACDecoder.swift:156-189 and
// LlamaTTS.swift:942-993.  Format: 8-byte LE header length, JSON header
// {name: {dtype, shape, data_offsets:[b,e]}, "__metadata__": {...}}, raw little-endian data.
#include "common.h"

#include <dirent.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <fstream>
#include <sstream>

static thread_local std::string g_last_error;

void mis_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
mis_status mis_fail(mis_status code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

extern "C" const char* mis_last_error(void) { return g_last_error.c_str(); }
extern "C" int mis_abi_version(void) { return MIS_ABI_VERSION; }
extern "C" void mis_free(void* p) {
    if (p) (void)hipHostFree(p);
}
extern "C" int mis_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------------------- JSON
namespace {
struct JsonParser {
    const std::string& s;
    size_t i = 0;
    explicit JsonParser(const std::string& t) : s(t) {}
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) i++; }
    [[noreturn]] void fail(const char* what) {
        throw MisError(MIS_ERR_INVALID_INPUT, std::string("json: ") + what + " at offset " + std::to_string(i));
    }
    JsonValue parse() {
        ws();
        if (i >= s.size()) fail("unexpected end");
        char c = s[i];
        JsonValue v;
        if (c == '{') {
            v.type = JsonValue::OBJ;
            i++;
            ws();
            if (i < s.size() && s[i] == '}') { i++; return v; }
            for (;;) {
                ws();
                if (i >= s.size() || s[i] != '"') fail("expected key");
                std::string k = str();
                ws();
                if (i >= s.size() || s[i] != ':') fail("expected ':'");
                i++;
                JsonValue val = parse();
                v.obj.emplace_back(std::move(k), std::move(val));
                ws();
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == '}') { i++; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            v.type = JsonValue::ARR;
            i++;
            ws();
            if (i < s.size() && s[i] == ']') { i++; return v; }
            for (;;) {
                v.arr.push_back(parse());
                ws();
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == ']') { i++; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v.type = JsonValue::STR;
            v.str = str();
        } else if (s.compare(i, 4, "true") == 0) { v.type = JsonValue::BOOL; v.b = true; i += 4; }
        else if (s.compare(i, 5, "false") == 0) { v.type = JsonValue::BOOL; v.b = false; i += 5; }
        else if (s.compare(i, 4, "null") == 0) { v.type = JsonValue::NUL; i += 4; }
        else {
            size_t j = i;
            while (j < s.size() && (isdigit((unsigned char)s[j]) || s[j] == '-' || s[j] == '+' || s[j] == '.' ||
                                    s[j] == 'e' || s[j] == 'E')) j++;
            if (j == i) fail("unexpected character");
            v.type = JsonValue::NUM;
            v.num = strtod(s.substr(i, j - i).c_str(), nullptr);
            i = j;
        }
        return v;
    }
    std::string str() {
        std::string out;
        i++;   // opening quote
        while (i < s.size() && s[i] != '"') {
            if (s[i] == '\\' && i + 1 < s.size()) {
                char e = s[i + 1];
                if (e == 'n') out += '\n';
                else if (e == 't') out += '\t';
                else if (e == 'u') { out += '?'; i += 4; }
                else out += e;
                i += 2;
            } else out += s[i++];
        }
        if (i >= s.size()) fail("unterminated string");
        i++;
        return out;
    }
};
}   // namespace

JsonValue json_parse(const std::string& text) {
    JsonParser p(text);
    return p.parse();
}

std::string read_text_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw MisError(MIS_ERR_NOT_INITIALIZED, "cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

// ---------------------------------------------------------------------------- safetensors
SafeTensorFile::~SafeTensorFile() {
    if (map) munmap(map, map_len);
}

void SafeTensorFile::open(const std::string& path) {
    int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw MisError(MIS_ERR_NOT_INITIALIZED, "cannot open " + path);
    struct stat st;
    if (fstat(fd, &st) != 0) { ::close(fd); throw MisError(MIS_ERR_NOT_INITIALIZED, "cannot stat " + path); }
    map_len = (size_t)st.st_size;
    map = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map == MAP_FAILED) { map = nullptr; throw MisError(MIS_ERR_NOT_INITIALIZED, "mmap failed for " + path); }
    if (map_len < 8) throw MisError(MIS_ERR_INVALID_INPUT, "safetensors file too small: " + path);
    const uint8_t* base = (const uint8_t*)map;
    uint64_t hlen = 0;
    memcpy(&hlen, base, 8);
    if (8 + hlen > map_len) throw MisError(MIS_ERR_INVALID_INPUT, "bad safetensors header length in " + path);
    std::string header((const char*)base + 8, (size_t)hlen);
    JsonValue root = json_parse(header);
    if (root.type != JsonValue::OBJ) throw MisError(MIS_ERR_INVALID_INPUT, "safetensors header is not an object");
    const uint8_t* data0 = base + 8 + hlen;
    size_t data_len = map_len - 8 - (size_t)hlen;
    for (auto& kv : root.obj) {
        if (kv.first == "__metadata__") continue;
        const JsonValue& t = kv.second;
        const JsonValue* dt = t.get("dtype");
        const JsonValue* sh = t.get("shape");
        const JsonValue* off = t.get("data_offsets");
        if (!dt || !sh || !off || off->arr.size() != 2)
            throw MisError(MIS_ERR_INVALID_INPUT, "malformed safetensors entry " + kv.first);
        SafeTensorEntry e;
        e.name = kv.first;
        e.dtype = dt->str;
        for (auto& d : sh->arr) e.shape.push_back((int64_t)d.num);
        size_t b = (size_t)off->arr[0].num, en = (size_t)off->arr[1].num;
        if (en < b || en > data_len) throw MisError(MIS_ERR_INVALID_INPUT, "safetensors offsets out of range: " + kv.first);
        e.data = data0 + b;
        e.nbytes = en - b;
        entries.push_back(std::move(e));
    }
}

std::vector<std::string> list_safetensors(const std::string& dir) {
    std::vector<std::string> out;
    DIR* d = opendir(dir.c_str());
    if (!d) throw MisError(MIS_ERR_NOT_INITIALIZED, "cannot open model directory " + dir);
    while (struct dirent* e = readdir(d)) {
        std::string n = e->d_name;
        if (n.size() > 12 && n.substr(n.size() - 12) == ".safetensors") out.push_back(dir + "/" + n);
    }
    closedir(d);
    std::sort(out.begin(), out.end());
    return out;
}

mis_dtype dtype_from_safetensors(const std::string& s) {
    if (s == "F32") return MIS_F32;
    if (s == "F16") return MIS_F16;
    if (s == "BF16") return MIS_BF16;
    if (s == "I32") return MIS_I32;
    throw MisError(MIS_ERR_INVALID_INPUT, "unsupported safetensors dtype " + s);
}
