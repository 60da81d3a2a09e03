// fuzz_readers -- the host's two file parsers (host/src/png.cpp, host/src/json.cpp) over mutated inputs, meant to be built
// with -fsanitize=address,undefined (make -C cart-slam_amd sanitize).  No GPU, no engine.
//   fuzz_readers <seed.png> <iterations> [<seed.json> ...]
// Every input must either parse or be refused with an exception; a crash, a sanitizer report or a hang fails the run.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include <zlib.h>

#include "cartslam_amd/json.hpp"
#include "cartslam_amd/png.hpp"

static std::vector<uint8_t> slurp(const char *path) {
    std::ifstream f(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

template <class Rng>
static std::vector<uint8_t> mutate(const std::vector<uint8_t> &seed, Rng &rng) {
    std::vector<uint8_t> v = seed;
    const int edits = 1 + (int)(rng() % 6);
    for (int e = 0; e < edits && !v.empty(); ++e) {
        const size_t at = rng() % v.size();
        switch (rng() % 6) {
            case 0: v[at] = (uint8_t)rng(); break;                                     // byte flip
            case 1: v[at] ^= (uint8_t)(1u << (rng() % 8)); break;                      // bit flip
            case 2: v.resize(at); break;                                               // truncate
            case 3: v.insert(v.begin() + at, (size_t)(rng() % 9), (uint8_t)rng()); break;   // insert a run
            case 4: if (at + 4 <= v.size()) { const uint32_t big = 0xfffffff0u + (uint32_t)(rng() % 16); std::memcpy(&v[at], &big, 4); } break;   // huge length / size field
            default: if (at + 1 < v.size()) v.erase(v.begin() + at, v.begin() + at + 1 + rng() % std::min<size_t>(16, v.size() - at - 1)); break;
        }
    }
    return v;
}

// a structurally valid PNG around RANDOM scanline data (random filter bytes, the invalid ones included): what a mutated file
// almost never reaches, because its zlib stream no longer inflates
template <class Rng>
static std::vector<uint8_t> randomPng(Rng &rng) {
    static const int ctypes[4] = {0, 2, 4, 6}, chans[4] = {1, 3, 2, 4};
    const int k = (int)(rng() % 4), w = 1 + (int)(rng() % 40), h = 1 + (int)(rng() % 24);
    std::vector<uint8_t> raw((size_t)(w * chans[k] + 1) * h);
    for (auto &b : raw) b = (uint8_t)rng();
    for (int y = 0; y < h; ++y) raw[(size_t)(w * chans[k] + 1) * y] = (uint8_t)(rng() % 6);   // filter types 0..4 and the invalid 5
    if (rng() % 5 == 0) raw.resize(raw.size() - 1 - rng() % raw.size() / 2);                  // too little data for the header's size
    std::vector<uint8_t> z(compressBound((uLong)raw.size()));
    uLongf zl = (uLongf)z.size();
    compress(z.data(), &zl, raw.data(), (uLong)raw.size());
    z.resize(zl);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    auto be = [&](uint32_t v) { for (int s = 24; s >= 0; s -= 8) out.push_back((uint8_t)(v >> s)); };
    auto chunk = [&](const char *tag, const std::vector<uint8_t> &body) {
        be((uint32_t)body.size()); out.insert(out.end(), tag, tag + 4); out.insert(out.end(), body.begin(), body.end()); be(0);   // the reader ignores the CRC
    };
    std::vector<uint8_t> ihdr;
    for (uint32_t v : {(uint32_t)w, (uint32_t)h}) for (int s = 24; s >= 0; s -= 8) ihdr.push_back((uint8_t)(v >> s));
    ihdr.insert(ihdr.end(), {8, (uint8_t)ctypes[k], 0, 0, 0});
    chunk("IHDR", ihdr); chunk("IDAT", z); chunk("IEND", {});
    return out;
}

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: fuzz_readers <seed.png> <iterations> [<seed.json> ...]\n"); return 2; }
    const std::vector<uint8_t> png = slurp(argv[1]);
    const int iters = std::atoi(argv[2]);
    if (png.empty()) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    std::mt19937_64 rng(0x5eedca27u);
    const std::string tmp = std::string(argv[1]) + ".fuzz.png";
    int parsed = 0, refused = 0;
    for (int i = 0; i < iters; ++i) {
        const std::vector<uint8_t> v = i == 0 ? png : (i & 1) ? mutate(png, rng) : randomPng(rng);
        std::ofstream(tmp, std::ios::binary).write(reinterpret_cast<const char *>(v.data()), (std::streamsize)v.size());
        try {
            cart::util::HostImage img;
            if (cart::util::readPng(tmp, img) && img.data.size() == (size_t)img.w * img.h * img.channels) ++parsed; else ++refused;
        } catch (const std::exception &) { ++refused; }
    }
    std::remove(tmp.c_str());
    std::printf("png: %d inputs, %d parsed, %d refused\n", iters, parsed, refused);
    if (parsed < 1) { std::printf("FAILED: the unmodified seed did not parse\n"); return 1; }
    // JSON: the seeds, their mutations, and hand-made nasties (deep nesting must be refused, not overflow the stack)
    std::vector<std::string> seeds = {"{\"a\": [1, 2.5e3, true, null, \"x\\n\"], \"b\": {\"c\": -0.5}}", std::string(100000, '['), std::string(5000, '{'), "\"\\", "{\"a\":", "[1,]", "1e999999", "-"};
    for (int a = 3; a < argc; ++a) { const auto b = slurp(argv[a]); seeds.emplace_back(b.begin(), b.end()); }
    int jparsed = 0, jrefused = 0;
    for (size_t s = 0; s < seeds.size(); ++s)
        for (int i = 0; i < (s < 8 ? 1 : iters); ++i) {
            std::string text = seeds[s];
            if (i > 0) { const auto m = mutate(std::vector<uint8_t>(text.begin(), text.end()), rng); text.assign(m.begin(), m.end()); }
            try { (void)cart::json::parse(text); ++jparsed; } catch (const std::exception &) { ++jrefused; }
        }
    std::printf("json: %d parsed, %d refused\n", jparsed, jrefused);
    std::printf("ok\n");
    return 0;
}
