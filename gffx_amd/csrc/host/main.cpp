// main.cpp -- `gffx` command line for the intersect path: `gffx index` (prerequisite),
// `gffx intersect` and `gffx depth` (BED source) (reference: main.rs:11-39, commands/depth.rs:34-72, commands/index.rs:11-23, commands/intersect.rs:32-70,
// utils/common.rs:17-52).  Flag names, short flags, defaults and groups follow the reference's
// clap derive; usage errors exit 2 like clap, run-time errors print `Error: <msg>` and exit 1.
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>

#include <thread>

#include "gffx.hpp"

namespace gffx {

namespace {

struct UsageError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct OptSpec {
    char short_name;  // 0 = none
    const char *long_name;
    bool takes_value;
};

// clap-style scan: -x VALUE, -xVALUE, -x=VALUE, --long VALUE, --long=VALUE, bundled -ev
std::map<std::string, std::vector<std::string>> parse_opts(int argc, char **argv, int first,
                                                           const std::vector<OptSpec> &specs) {
    std::map<std::string, std::vector<std::string>> got;
    auto find_long = [&](const std::string &n) -> const OptSpec * {
        for (const auto &s : specs)
            if (n == s.long_name) return &s;
        return nullptr;
    };
    auto find_short = [&](char c) -> const OptSpec * {
        for (const auto &s : specs)
            if (s.short_name && s.short_name == c) return &s;
        return nullptr;
    };
    for (int i = first; i < argc; ++i) {
        const std::string a = argv[i];
        if (a.rfind("--", 0) == 0 && a.size() > 2) {
            const size_t eq = a.find('=');
            const std::string name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            const OptSpec *s = find_long(name);
            if (!s) throw UsageError("unexpected argument '--" + name + "' found");
            if (s->takes_value) {
                if (eq != std::string::npos) {
                    got[s->long_name].push_back(a.substr(eq + 1));
                } else {
                    if (i + 1 >= argc) throw UsageError("a value is required for '--" + name + "' but none was supplied");
                    got[s->long_name].push_back(argv[++i]);
                }
            } else {
                if (eq != std::string::npos) throw UsageError("unexpected value for '--" + name + "'");
                got[s->long_name].push_back("");
            }
        } else if (a.size() >= 2 && a[0] == '-' && a != "--") {
            for (size_t k = 1; k < a.size(); ++k) {
                const OptSpec *s = find_short(a[k]);
                if (!s) throw UsageError(std::string("unexpected argument '-") + a[k] + "' found");
                if (s->takes_value) {
                    std::string v = a.substr(k + 1);
                    if (!v.empty() && v[0] == '=') v.erase(0, 1);
                    if (v.empty()) {
                        if (i + 1 >= argc)
                            throw UsageError(std::string("a value is required for '-") + a[k] + "' but none was supplied");
                        v = argv[++i];
                    }
                    got[s->long_name].push_back(v);
                    break;
                }
                got[s->long_name].push_back("");
            }
        } else {
            throw UsageError("unexpected argument '" + a + "' found");
        }
    }
    for (const auto &[name, vals] : got)
        if (vals.size() > 1)
            throw UsageError("the argument '--" + name + "' cannot be used multiple times");
    return got;
}

const char *kTopUsage =
    "Usage: gffx <COMMAND>\n\nCommands:\n"
    "  index      Build index for GFF file\n"
    "  intersect  Extract models by a region or regions from a BED file (MI355X engine)\n"
    "  depth      Compute coverage depth across genomic features from a BED file (MI355X engine)\n"
    "  coverage   Compute coverage breadth across genomic features from a BED file (MI355X engine)\n"
    "  help       Print this message\n";

const char *kIntersectUsage =
    "Usage: gffx intersect [OPTIONS] --input <FILE> <--region <REGION>|--bed <BED>>\n\nOptions:\n"
    "  -i, --input <FILE>       Input GFF file path\n"
    "  -o, --output <FILE>      Output file (stdout if not provided)\n"
    "  -e, --entire_group       Return the entire feature group for each match\n"
    "  -T, --types <TYPES>      Comma-separated feature types to retain (e.g. exon,gene)\n"
    "  -t, --threads <NUM>      Number of threads for parallel processing [default: 12]\n"
    "  -v, --verbose            Enable verbose output\n"
    "  -r, --region <REGION>    Single region in format \"chr:start-end\"\n"
    "  -b, --bed <BED>          BED file containing regions\n"
    "  -c, --contained          Only return features fully contained within regions\n"
    "  -C, --contains-region    Only return features that fully contain the regions\n"
    "  -O, --overlap            Return any overlapping features (default)\n"
    "  -I, --invert             Invert the selection (exclude matching features)\n"
    "      --device <N>         HIP device to run on [default: 0]\n"
    "      --gpus <N>           Shard the BED regions by chromosome bucket over N devices [default: 1]\n"
    "      --stats-json <FILE>  Write the run's stage timers and counts as one JSON object\n";

const char *kDepthUsage =
    "Usage: gffx depth [OPTIONS] --input <FILE> --source <SOURCE>\n\nOptions:\n"
    "  -i, --input <FILE>           Input GFF file path\n"
    "  -s, --source <SOURCE>        Input source (BED; BAM/SAM/CRAM are not supported by this build)\n"
    "  -o, --output <FILE>          Output file (stdout if not provided)\n"
    "      --bin-shift <BIN_SHIFT>  Bin width parameter (2^k bp) [default: 12]\n"
    "  -t, --threads <THREADS>      Number of threads for parallel processing [default: 12]\n"
    "  -v, --verbose                Enable verbose output\n"
    "      --device <N>             HIP device to run on [default: 0]\n"
    "      --gpus <N>               Spread the BED rows over N devices [default: 1]\n";

const char *kCoverageUsage =
    "Usage: gffx coverage [OPTIONS] --input <FILE> --source <SOURCE>\n\nOptions:\n"
    "  -i, --input <FILE>       Input GFF file path\n"
    "  -s, --source <SOURCE>    Input source (BED; BAM/SAM/CRAM are not supported by this build)\n"
    "  -o, --output <FILE>      Output file (stdout if not provided)\n"
    "  -t, --threads <NUM>      Number of threads for parallel processing [default: 12]\n"
    "  -v, --verbose            Enable verbose output\n"
    "      --device <N>         HIP device to run on [default: 0]\n";

const char *kIndexUsage =
    "Usage: gffx index [OPTIONS] --input <INPUT>\n\nOptions:\n"
    "  -i, --input <INPUT>\n"
    "  -a, --attribute <ATTRIBUTE>    [default: gene_name]\n"
    "  -s, --skip-types <SKIP_TYPES>  [default: remark,note,comment,region,gap,assembly_gap,contig,scaffold,source]\n"
    "  -v, --verbose\n";

size_t parse_size(const std::string &v, const char *flag) {
    const auto x = parse_u32_rust(v);
    if (!x) throw UsageError(std::string("invalid value '") + v + "' for '" + flag + "'");
    return *x;
}

int run_intersect_cli(int argc, char **argv) {
    static const std::vector<OptSpec> specs = {
        {'i', "input", true},   {'o', "output", true},     {'e', "entire_group", false}, {'T', "types", true},
        {'t', "threads", true}, {'v', "verbose", false},   {'r', "region", true},        {'b', "bed", true},
        {'c', "contained", false}, {'C', "contains-region", false}, {'O', "overlap", false},
        {'I', "invert", false}, {0, "device", true},       {0, "gpus", true},           {0, "stats-json", true},
        {'h', "help", false}};
    const auto o = parse_opts(argc, argv, 2, specs);
    if (o.count("help")) {
        std::fputs(kIntersectUsage, stdout);
        return 0;
    }
    commands::intersect::IntersectArgs a;
    if (!o.count("input")) throw UsageError("the following required arguments were not provided:\n  --input <FILE>");
    a.common.input = o.at("input")[0];
    if (o.count("output")) a.common.output = o.at("output")[0];
    a.common.entire_group = o.count("entire_group") > 0;
    if (o.count("types")) a.common.types = o.at("types")[0];
    if (o.count("threads")) a.common.threads = parse_size(o.at("threads")[0], "--threads <NUM>");
    a.common.verbose = o.count("verbose") > 0;
    if (o.count("region")) a.region = o.at("region")[0];
    if (o.count("bed")) a.bed = o.at("bed")[0];
    // ArgGroup "regions": required, exactly one (intersect.rs:37-39)
    if (a.region && a.bed)
        throw UsageError("the argument '--region <REGION>' cannot be used with '--bed <BED>'");
    if (!a.region && !a.bed)
        throw UsageError("the following required arguments were not provided:\n  <--region <REGION>|--bed <BED>>");
    a.contained = o.count("contained") > 0;
    a.contains_region = o.count("contains-region") > 0;
    a.overlap = o.count("overlap") > 0;
    if (a.contained + a.contains_region + a.overlap > 1)  // ArgGroup "mode" (intersect.rs:40-42)
        throw UsageError("the arguments '--contained', '--contains-region' and '--overlap' cannot be used together");
    a.invert = o.count("invert") > 0;
    if (o.count("device")) a.device = static_cast<int>(parse_size(o.at("device")[0], "--device <N>"));
    if (o.count("gpus")) a.gpus = static_cast<int>(std::max<size_t>(1, std::min<size_t>(64, parse_size(o.at("gpus")[0], "--gpus <N>"))));
    if (o.count("stats-json")) g_run_stats.path = o.at("stats-json")[0];
    commands::intersect::run(a);
    return 0;
}

int run_depth_cli(int argc, char **argv) {
    static const std::vector<OptSpec> specs = {{'i', "input", true},   {'s', "source", true},   {'o', "output", true},
                                               {0, "bin-shift", true}, {'t', "threads", true},  {'v', "verbose", false},
                                               {0, "device", true},    {0, "gpus", true},       {0, "stats-json", true}, {'h', "help", false}};
    const auto o = parse_opts(argc, argv, 2, specs);
    if (o.count("help")) {
        std::fputs(kDepthUsage, stdout);
        return 0;
    }
    commands::depth::DepthArgs a;
    if (!o.count("input") || !o.count("source"))
        throw UsageError(std::string("the following required arguments were not provided:") +
                         (o.count("input") ? "" : "\n  --input <FILE>") + (o.count("source") ? "" : "\n  --source <SOURCE>"));
    a.input = o.at("input")[0];
    a.source = o.at("source")[0];
    if (o.count("output")) a.output = o.at("output")[0];
    if (o.count("bin-shift")) a.bin_shift = static_cast<uint32_t>(parse_size(o.at("bin-shift")[0], "--bin-shift <BIN_SHIFT>"));
    if (o.count("threads")) a.threads = parse_size(o.at("threads")[0], "--threads <THREADS>");
    a.verbose = o.count("verbose") > 0;
    if (o.count("device")) a.device = static_cast<int>(parse_size(o.at("device")[0], "--device <N>"));
    if (o.count("gpus")) a.gpus = static_cast<int>(std::max<size_t>(1, std::min<size_t>(64, parse_size(o.at("gpus")[0], "--gpus <N>"))));
    if (o.count("stats-json")) g_run_stats.path = o.at("stats-json")[0];
    commands::depth::run(a);
    return 0;
}

int run_coverage_cli(int argc, char **argv) {
    static const std::vector<OptSpec> specs = {{'i', "input", true},   {'s', "source", true},   {'o', "output", true},
                                               {'t', "threads", true}, {'v', "verbose", false}, {0, "device", true},
                                               {0, "stats-json", true}, {'h', "help", false}};
    const auto o = parse_opts(argc, argv, 2, specs);
    if (o.count("help")) {
        std::fputs(kCoverageUsage, stdout);
        return 0;
    }
    commands::coverage::CoverageArgs a;
    if (!o.count("input") || !o.count("source"))
        throw UsageError(std::string("the following required arguments were not provided:") +
                         (o.count("input") ? "" : "\n  --input <FILE>") + (o.count("source") ? "" : "\n  --source <SOURCE>"));
    a.input = o.at("input")[0];
    a.source = o.at("source")[0];
    if (o.count("output")) a.output = o.at("output")[0];
    if (o.count("threads")) a.threads = parse_size(o.at("threads")[0], "--threads <NUM>");
    a.verbose = o.count("verbose") > 0;
    if (o.count("device")) a.device = static_cast<int>(parse_size(o.at("device")[0], "--device <N>"));
    if (o.count("stats-json")) g_run_stats.path = o.at("stats-json")[0];
    commands::coverage::run(a);
    return 0;
}

int run_index_cli(int argc, char **argv) {
    static const std::vector<OptSpec> specs = {{'i', "input", true},
                                               {'a', "attribute", true},
                                               {'s', "skip-types", true},
                                               {'v', "verbose", false},
                                               {'h', "help", false}};
    const auto o = parse_opts(argc, argv, 2, specs);
    if (o.count("help")) {
        std::fputs(kIndexUsage, stdout);
        return 0;
    }
    if (!o.count("input")) throw UsageError("the following required arguments were not provided:\n  --input <INPUT>");
    const std::string input = o.at("input")[0];
    const std::string attr = o.count("attribute") ? o.at("attribute")[0] : "gene_name";  // commands/index.rs:15
    const std::string skip = o.count("skip-types")
                                 ? o.at("skip-types")[0]
                                 : "remark,note,comment,region,gap,assembly_gap,contig,scaffold,source";  // :18
    const bool verbose = o.count("verbose") > 0;
    if (verbose) std::printf("Indexing: %s\n", input.c_str());  // commands/index.rs:26-28
    build_index(input, attr, skip, verbose);
    // addition: the all-line SoA image `<gff>.lsoa` for depth / coverage (block_table.cpp); GFFX_LINE_TABLE=off skips it
    const char *lt = std::getenv("GFFX_LINE_TABLE");
    if (!(lt && std::string(lt) == "off")) {
        const index_loader::GofMap gof = index_loader::load_gof(input);
        const MappedFile text(input);
        unsigned hw = std::thread::hardware_concurrency();
        // Both images are optional accelerators (the commands fall back to the text walk without them): a failure to write one
        // -- a full or read-only directory, a quota -- is a warning, the partial file is removed, and the index the reference
        // defines (the eight side-cars above) stands.
        try {
            const auto t = commands::depth::build_block_table(gof, text.view(), std::min(hw ? hw : 1u, 12u));
            commands::depth::write_block_table(append_suffix(input, ".lsoa"), t, text.size(), commands::depth::line_table_key(input, gof));
            if (verbose) std::printf("Line table image: %zu lines in %zu blocks.\n", t.line_start.size(), t.block_line_off.size() - 1);
        } catch (const Error &e) {
            std::fprintf(stderr, "[WARN] line table image %s not written: %s\n", append_suffix(input, ".lsoa").c_str(), e.what());
            std::remove(append_suffix(input, ".lsoa").c_str());
        }
        // ... and the all-line table `<gff>.lall` of intersect's per-line mode (line_index.cpp)
        try {
            const auto all = commands::intersect::build_all_lines(text.view(), std::min(hw ? hw : 1u, 12u));
            commands::intersect::write_all_lines(append_suffix(input, ".lall"), all, text.size(), commands::depth::line_table_key(input, gof));
            if (verbose) std::printf("All-line table: %zu lines, %zu seqid names, %zu types.\n", all.ls.size(), all.seq_names.size(), all.type_names.size());
        } catch (const Error &e) {
            std::fprintf(stderr, "[WARN] all-line table %s not written: %s\n", append_suffix(input, ".lall").c_str(), e.what());
            std::remove(append_suffix(input, ".lall").c_str());
        }
    } else {
        std::remove(append_suffix(input, ".lsoa").c_str());  // an image of an earlier index run would be stale
        std::remove(append_suffix(input, ".lall").c_str());
    }
    if (verbose) std::printf("Index created successfully.\n");
    return 0;
}

}  // namespace

int cli_main(int argc, char **argv) {
    const char *usage = kTopUsage;
    try {
        if (argc < 2) throw UsageError("a subcommand is required");
        const std::string cmd = argv[1];
        if (cmd == "help" || cmd == "--help" || cmd == "-h") {
            std::fputs(kTopUsage, stdout);
            return 0;
        }
        if (cmd == "intersect") {
            usage = kIntersectUsage;
            return run_intersect_cli(argc, argv);
        }
        if (cmd == "index") {
            usage = kIndexUsage;
            return run_index_cli(argc, argv);
        }
        if (cmd == "depth") {
            usage = kDepthUsage;
            return run_depth_cli(argc, argv);
        }
        if (cmd == "coverage") {
            usage = kCoverageUsage;
            return run_coverage_cli(argc, argv);
        }
        throw UsageError("unrecognized subcommand '" + cmd + "' (this build carries the intersect, depth and coverage paths only)");
    } catch (const UsageError &e) {
        std::fprintf(stderr, "error: %s\n\n%s\nFor more information, try '--help'.\n", e.what(), usage);
        return 2;
    } catch (const Error &e) {
        std::fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
}

}  // namespace gffx

#ifndef GFFX_NO_MAIN
// The process ends the ordinary way: destructors, atexit handlers, HIP runtime shutdown.  GFFX_EXIT=fast (opt-in, for
// throughput runs on multi-GB inputs) skips that teardown once cli_main has closed and checked every output it wrote --
// unmapping a 2.4 GB BED file, freeing the region stores and shutting HIP down cost ~0.1 s of a 0.9 s run there, and the
// kernel reclaims all of it faster when the process simply ends.  Never under a preloaded tool (a profiler flushes its
// trace from an exit handler).
int main(int argc, char **argv) {
    const int rc = gffx::cli_main(argc, argv);
    const char *how = std::getenv("GFFX_EXIT");
    if (!(how && std::string(how) == "fast")) return rc;
    const char *preload = std::getenv("LD_PRELOAD");
    if ((preload && *preload) || std::getenv("ROCP_TOOL_LIBRARIES") || std::getenv("ROCPROFILER_REGISTER_ROOT") || std::getenv("HSA_TOOLS_LIB"))
        return rc;
    std::fflush(stdout);
    std::fflush(stderr);
    _exit(rc);
}
#endif
