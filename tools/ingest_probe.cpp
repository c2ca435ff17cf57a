// Where does the PAF ingest spend its time?  read_paf_parallel on one file, with an empty name
// table (tokenising + number parsing + column stores only) and with the real one (plus look-ups).
//   g++ -O2 -std=c++17 -Irala_amd/host -o tools/_bin/ingest_probe tools/ingest_probe.cpp rala_amd/host/io.cpp -lz -pthread
//   tools/_bin/ingest_probe <file.paf> <n_reads> <threads>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <string>
#include <vector>

#include "io.hpp"

using namespace rala::io;

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    if (argc < 4) return 1;
    const char* path = argv[1];
    const size_t n = (size_t)atoll(argv[2]);
    const unsigned threads = (unsigned)atoi(argv[3]);
    std::vector<std::string> names(n);
    for (size_t i = 0; i < n; ++i) names[i] = "r" + std::to_string(i);
    std::vector<uint32_t> len(n, 10000);
    for (int variant = 0; variant < 2; ++variant) {
        NameTable tab;
        if (variant == 1) tab.build(names);
        for (int rep = 0; rep < 3; ++rep) {
            const double t0 = now();
            size_t lines = 0;
            {
                OverlapColumns c;
                int64_t bad = -1;
                read_paf_parallel(path, tab, len, false, threads, c, &bad);
                lines = c.size();
                const double t1 = now();
                printf("%s: %zu lines, %.0f ms, %.0f ns per line and thread", variant ? "real table " : "empty table", lines,
                       (t1 - t0) * 1e3, (t1 - t0) * 1e9 * threads / lines);
            }
            printf(" (+ %.0f ms to free the columns)\n", (now() - t0) * 1e3 - 0);
        }
    }
    return 0;
}
