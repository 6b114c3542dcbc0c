// ThreadSanitizer harness for the progressive PNG encoder (pngwriter::Progressive): the source buffer is filled band by band while the stripe
// workers read the rows already declared ready; the file must be the one-shot encoder's.  Built and run by tools/run_cpu_sanitizers.sh.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdint>
extern "C" {
int mcu_png_encode_storage(const float*, uint32_t, uint32_t, float, int, int, uint8_t**, size_t*);
int mcu_png_encode_progressive(const void*, uint32_t, uint32_t, float, int, int, int, uint8_t**, size_t*);
int mcu_png_progressive_abandon(uint32_t, uint32_t, int, int, uint32_t);
void mcu_free(void*);
}
int main() {
    const uint32_t w = 900, h = 700;
    std::vector<float> img((size_t)w * h * 4);
    unsigned s = 12345;
    for (auto& v : img) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / 16777216.0f * 300.0f - 20.0f; }
    int bad = 0;
    for (int threads : {1, 4, 8}) {
        uint8_t* a = nullptr; size_t na = 0;
        if (mcu_png_encode_storage(img.data(), w, h, 1.0f, 0, threads, &a, &na)) return 2;
        for (int bands : {1, 3, 11, 700}) {
            uint8_t* b = nullptr; size_t nb = 0;
            if (mcu_png_encode_progressive(img.data(), w, h, 1.0f, 0, threads, bands, &b, &nb)) return 3;
            if (na != nb || std::memcmp(a, b, na)) bad++;
            mcu_free(b);
        }
        mcu_free(a);
    }
    // an image abandoned half-way (the cancel / wake-up / join of the workers, both routes)
    for (int threads : {1, 4, 8})
        for (uint32_t ready : {0u, 1u, 350u, 700u})
            for (int route : {0, 1})
                if (mcu_png_progressive_abandon(w, h, route, threads, ready)) bad++;
    printf("mismatches %d\n", bad);
    return bad != 0;
}
