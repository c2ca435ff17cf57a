// `rala` command line of the MI355X build.
//
// The option set and the usage text are the interface of rvaser/rala (src/main.cpp:11-20,
// :99-127) and are kept so that scripts such as misc/raven.sh keep working; everything else
// here is this build's own driver: settings are parsed into a plain struct, the work is a
// short list of stages, FASTA goes out through one buffered writer.  Additions: `--gpus`
// (devices of one node for the pile stage) and the RALA_GPUS environment variable.
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <string>
#include <vector>

#include "graph.hpp"
#include "sequence.hpp"

namespace {

const char* const kVersion = "v1.0.0-mi355x";

struct Settings {
    std::string sequences, overlaps;     // positional
    std::string sensitive_overlaps;      // -s
    std::string debug_prefix;            // -d
    uint32_t threads = 1;                // -t
    uint32_t gpus = 1;                   // --gpus / RALA_GPUS
    bool first_pass_only = false;        // -p: print the uncontained reads and stop
    bool keep_unassembled = false;       // -u
};

enum class Parsed { kRun, kDone, kBad };

const option kLongOptions[] = {
    {"preconstruct", no_argument, nullptr, 'p'},
    {"include-unassembled", no_argument, nullptr, 'u'},
    {"debug", required_argument, nullptr, 'd'},
    {"sensitive-overlaps", required_argument, nullptr, 's'},
    {"threads", required_argument, nullptr, 't'},
    {"gpus", required_argument, nullptr, 'g'},
    {"version", no_argument, nullptr, 'v'},
    {"help", no_argument, nullptr, 'h'},
    {nullptr, 0, nullptr, 0}};

void usage(FILE* to) {
    static const char* const text =
        "usage: rala [options ...] <sequences> <overlaps>\n"
        "\n"
        "    <sequences>\n"
        "        input file in FASTA/FASTQ format (can be compressed with gzip)\n"
        "        containing sequences\n"
        "    <overlaps>\n"
        "        input file in MHAP/PAF format (can be compressed with gzip)\n"
        "        containing pairwise overlaps\n"
        "\n"
        "    options:\n"
        "        -p, --preconstruct\n"
        "            print uncontained sequences for second iteration\n"
        "        -s, --sensitive-overlaps <file>\n"
        "            input file in MHAP/PAF format (can be compress with gzip)\n"
        "            containing more sensitive overlaps\n"
        "        -u, --include-unassembled\n"
        "            output unassembled sequences (singletons and short contigs)\n"
        "        -d, --debug <string>\n"
        "            enable debug output with given prefix\n"
        "        -t, --threads <int>\n"
        "            default: 1\n"
        "            number of threads (host side: readers, clean-up stages)\n"
        "        --gpus <int>\n"
        "            default: 1 (or RALA_GPUS)\n"
        "            number of MI355X devices of this node for the pile stage\n"
        "        --version\n"
        "            prints the version number\n"
        "        -h, --help\n"
        "            prints the usage\n";
    fputs(text, to);
}

bool to_count(const char* text, uint32_t& out) {
    char* end = nullptr;
    const long v = strtol(text, &end, 10);
    if (end == text || *end != '\0' || v < 0 || v > 1 << 20) return false;
    out = (uint32_t)v;
    return true;
}

Parsed parse(int argc, char** argv, Settings& st) {
    if (const char* env = getenv("RALA_GPUS")) {
        if (!to_count(env, st.gpus) || st.gpus == 0) st.gpus = 1;
    }
    for (;;) {
        const int c = getopt_long(argc, argv, "pud:s:t:h", kLongOptions, nullptr);
        if (c < 0) break;
        if (c == 'p') st.first_pass_only = true;
        else if (c == 'u') st.keep_unassembled = true;
        else if (c == 'd') st.debug_prefix = optarg;
        else if (c == 's') st.sensitive_overlaps = optarg;
        else if (c == 't') st.threads = (uint32_t)atoi(optarg);       // atoi like the reference (main.cpp:48)
        else if (c == 'g') {
            if (!to_count(optarg, st.gpus) || st.gpus == 0) {
                fprintf(stderr, "[rala::] error: --gpus needs a positive number!\n");
                return Parsed::kBad;
            }
        } else if (c == 'v') {
            printf("%s\n", kVersion);
            return Parsed::kDone;
        } else if (c == 'h') {
            usage(stdout);
            return Parsed::kDone;
        } else {
            return Parsed::kBad;
        }
    }
    if (argc - optind < 2) {
        fprintf(stderr, "[rala::] error: missing input file(s)!\n");
        usage(stdout);
        return Parsed::kBad;
    }
    st.sequences = argv[optind];
    st.overlaps = argv[optind + 1];
    return Parsed::kRun;
}

// FASTA records to stdout through one large buffer (contigs are megabases long)
class FastaWriter {
public:
    explicit FastaWriter(FILE* to) : to_(to) {
        static char buffer[1u << 20];               // lives as long as the stream may use it
        setvbuf(to_, buffer, _IOFBF, sizeof(buffer));
    }
    ~FastaWriter() { fflush(to_); }
    void write(const std::vector<std::unique_ptr<rala::Sequence>>& records) {
        for (const auto& r : records) {
            fputc('>', to_);
            fwrite(r->name().data(), 1, r->name().size(), to_);
            fputc('\n', to_);
            fwrite(r->data().data(), 1, r->data().size(), to_);
            fputc('\n', to_);
        }
    }

private:
    FILE* to_;
};

int run(const Settings& st) {
    {
        // rala::Graph reads it when it opens its devices; st.gpus is the environment's value unless
        // --gpus was given, which wins (also --gpus 1 over RALA_GPUS=4)
        const std::string n = std::to_string(st.gpus);
        setenv("RALA_GPUS", n.c_str(), 1);
    }
    std::unique_ptr<rala::Graph> graph = rala::createGraph(st.sequences, st.overlaps, st.threads);
    graph->construct(st.sensitive_overlaps);

    std::vector<std::unique_ptr<rala::Sequence>> out;
    if (st.first_pass_only) {
        graph->extract_nodes(out);
    } else {
        graph->simplify();
        graph->print_debug(st.debug_prefix);
        graph->extract_contigs(out, !st.keep_unassembled);
    }
    FastaWriter(stdout).write(out);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    Settings st;
    switch (parse(argc, argv, st)) {
        case Parsed::kDone: return 0;
        case Parsed::kBad: return 1;
        case Parsed::kRun: break;
    }
    // (read by the HIP runtime when it starts - nothing has touched the device yet: eight hardware queues instead of four for a
    // run over several GPUs - the streams of a rank's two contexts and RCCL's share them, and with four some kernels meant to run
    // beside each other run behind each other)
    if (st.gpus > 1) setenv("GPU_MAX_HW_QUEUES", "8", 0);
    return run(st);
}
