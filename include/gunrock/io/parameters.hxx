// parameters.hxx -- command line of the example drivers.
// API parity: include/gunrock/io/parameters.hxx:16-291 (reference): parameters_t
// (argc, argv, "<Algorithm Name>") with public fields filename, source_string,
// json_dir, json_file, tag_string, num_runs, export_metrics, validate, binary and
// the operator knobs; get_options(); parse_source_string, parse_tag_string,
// parse_load_balance, parse_filter_algorithm, parse_uniquify_algorithm.
// The reference parses with cxxopts v3 (not vendored, fetched at configure time);
// this is a self-contained parser with the same surface: long options as
// `--name value` or `--name=value`, short aliases -m -s -n -d -f -t, boolean
// switches, `--help`, unknown option => exception, missing --market => help + exit(0).
// --src/--num_runs/--validate are registered per algorithm name as upstream.
#pragma once

#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <map>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include <gunrock/algorithms/algorithms.hxx>
#include <gunrock/error.hxx>
#include <gunrock/util/filepath.hxx>

namespace gunrock {
namespace io {
namespace cli {

inline operators::load_balance_t parse_load_balance(std::string str) {
  std::transform(str.begin(), str.end(), str.begin(), ::tolower);
  static const std::map<std::string, operators::load_balance_t> names = {
      {"thread_mapped", operators::load_balance_t::thread_mapped},
      {"warp_mapped", operators::load_balance_t::warp_mapped},
      {"block_mapped", operators::load_balance_t::block_mapped},
      {"bucketing", operators::load_balance_t::bucketing},
      {"merge_path", operators::load_balance_t::merge_path},
      {"merge_path_v2", operators::load_balance_t::merge_path_v2},
      {"work_stealing", operators::load_balance_t::work_stealing}};
  auto it = names.find(str);
  return it == names.end() ? operators::load_balance_t::block_mapped : it->second;  // unknown => default
}

inline operators::filter_algorithm_t parse_filter_algorithm(std::string str) {
  std::transform(str.begin(), str.end(), str.begin(), ::tolower);
  if (str == "remove") return operators::filter_algorithm_t::remove;
  if (str == "compact") return operators::filter_algorithm_t::compact;
  if (str == "bypass") return operators::filter_algorithm_t::bypass;
  return operators::filter_algorithm_t::predicated;
}

inline operators::uniquify_algorithm_t parse_uniquify_algorithm(std::string str) {
  std::transform(str.begin(), str.end(), str.begin(), ::tolower);
  return str == "unique_copy" ? operators::uniquify_algorithm_t::unique_copy
                              : operators::uniquify_algorithm_t::unique;
}

struct parameters_t {
  std::string filename;
  std::string source_string = "";
  std::string json_dir = ".";
  std::string json_file = "";
  std::string tag_string = "";
  int num_runs = 1;
  bool export_metrics = false;
  bool validate = false;
  bool binary = false;

  operators::load_balance_t advance_load_balance = operators::load_balance_t::block_mapped;
  operators::filter_algorithm_t filter_algorithm = operators::filter_algorithm_t::predicated;
  bool enable_filter = false;
  bool enable_uniquify = false;
  operators::uniquify_algorithm_t uniquify_algorithm = operators::uniquify_algorithm_t::unique;
  bool best_effort_uniquify = true;
  float uniquify_percent = 100.0f;
  int engine_flags = 0;  // --generic_operators sets bit 0
  operators::advance_direction_t advance_direction = operators::advance_direction_t::forward;

  parameters_t(int argc, char** argv, std::string algorithm) {
    struct spec_t { bool takes_value; std::string help; };
    std::map<std::string, spec_t> known = {
        {"help", {false, "Print help"}},
        {"export_metrics", {false, "export performance analysis metrics"}},
        {"market", {true, "Matrix file"}},
        {"json_dir", {true, "JSON output directory"}},
        {"json_file", {true, "JSON output file"}},
        {"tag", {true, "Tags for the JSON output; comma-separated string of tags"}},
        {"advance_load_balance", {true, "Load balancing technique for advance operator (thread_mapped, "
                                        "warp_mapped, block_mapped, merge_path, ...)"}},
        {"filter_algorithm", {true, "Filter algorithm (remove, predicated, compact, bypass)"}},
        {"enable_filter", {false, "Enable filter operator"}},
        {"enable_uniquify", {false, "Enable uniquify operator"}},
        {"uniquify_algorithm", {true, "Uniquify algorithm (unique, unique_copy)"}},
        {"best_effort_uniquify", {false, "Best-effort uniquification (skip sorting)"}},
        {"uniquify_percent", {true, "Percentage of elements to uniquify (0-100)"}},
        {"generic_operators", {false, "Run the generic operator pipeline instead of the fused engine"}},
        {"advance_direction", {true, "Advance direction (forward, optimized = direction-optimising BFS)"}},
        {"num_runs", {true, "Number of runs"}}};
    std::map<char, std::string> shorts = {{'m', "market"}, {'d', "json_dir"}, {'f', "json_file"},
                                          {'t', "tag"},    {'n', "num_runs"}};
    const bool sourced = algorithm == "Betweenness Centrality" || algorithm == "Breadth First Search" ||
                         algorithm == "Single Source Shortest Path";
    if (sourced) {
      known["src"] = {true, "Source(s) (random if omitted); comma-separated string of ints"};
      shorts['s'] = "src";
      if (algorithm != "Betweenness Centrality") known["validate"] = {false, "CPU validation"};
    }

    auto help = [&]() {
      std::cout << algorithm << " example\nUsage:\n  " << (argc > 0 ? argv[0] : "driver") << " [OPTION...]\n\n";
      for (auto& kv : known) {
        std::string sh;
        for (auto& s : shorts)
          if (s.second == kv.first) sh = std::string("-") + s.first + ", ";
        std::cout << "  " << sh << "--" << kv.first << (kv.second.takes_value ? " arg" : "") << "  "
                  << kv.second.help << "\n";
      }
      std::cout << std::endl;
    };

    std::map<std::string, std::string> seen;
    for (int i = 1; i < argc; ++i) {
      std::string tok = argv[i], name, value;
      bool has_value = false;
      if (tok.rfind("--", 0) == 0) {
        name = tok.substr(2);
        const auto eq = name.find('=');
        if (eq != std::string::npos) {
          value = name.substr(eq + 1);
          name = name.substr(0, eq);
          has_value = true;
        }
      } else if (tok.size() >= 2 && tok[0] == '-') {
        auto it = shorts.find(tok[1]);
        error::throw_if_exception(it == shorts.end(), "Option '" + tok + "' does not exist");
        name = it->second;
        if (tok.size() > 2) {
          value = tok.substr(2);
          has_value = true;
        }
      } else {
        error::throw_if_exception(true, "Unexpected positional argument '" + tok + "'");
      }
      auto spec = known.find(name);
      error::throw_if_exception(spec == known.end(), "Option '" + name + "' does not exist");
      if (spec->second.takes_value && !has_value) {
        error::throw_if_exception(i + 1 >= argc, "Option '" + name + "' is missing an argument");
        value = argv[++i];
      }
      seen[name] = value;
    }

    if (seen.count("help") || !seen.count("market")) {
      help();
      std::exit(0);
    }
    filename = seen["market"];
    if (util::is_binary_csr(filename)) {
      binary = true;
    } else if (!util::is_market(filename)) {
      help();
      std::exit(0);
    }
    validate = seen.count("validate") > 0;
    export_metrics = seen.count("export_metrics") > 0;
    if (seen.count("num_runs")) num_runs = std::stoi(seen["num_runs"]);
    if (seen.count("tag")) tag_string = seen["tag"];
    if (seen.count("src")) source_string = seen["src"];
    if (seen.count("json_dir")) json_dir = seen["json_dir"];
    if (seen.count("json_file")) json_file = seen["json_file"];
    if (seen.count("advance_load_balance")) advance_load_balance = parse_load_balance(seen["advance_load_balance"]);
    if (seen.count("filter_algorithm")) filter_algorithm = parse_filter_algorithm(seen["filter_algorithm"]);
    enable_filter = seen.count("enable_filter") > 0;
    enable_uniquify = seen.count("enable_uniquify") > 0;
    if (seen.count("uniquify_algorithm")) uniquify_algorithm = parse_uniquify_algorithm(seen["uniquify_algorithm"]);
    if (seen.count("best_effort_uniquify")) best_effort_uniquify = true;
    if (seen.count("uniquify_percent")) uniquify_percent = std::stof(seen["uniquify_percent"]);
    if (seen.count("generic_operators")) engine_flags |= 1;
    if (seen.count("advance_direction")) {
      std::string d = seen["advance_direction"];
      std::transform(d.begin(), d.end(), d.begin(), ::tolower);
      advance_direction = d == "optimized" ? operators::advance_direction_t::optimized
                          : d == "backward" ? operators::advance_direction_t::backward
                                            : operators::advance_direction_t::forward;
    }
  }

  gunrock::options_t get_options() const {
    gunrock::options_t o;
    o.advance_load_balance = advance_load_balance;
    o.filter_algorithm = filter_algorithm;
    o.enable_filter = enable_filter;
    o.enable_uniquify = enable_uniquify;
    o.uniquify_algorithm = uniquify_algorithm;
    o.best_effort_uniquify = best_effort_uniquify;
    o.uniquify_percent = uniquify_percent;
    o.engine_flags = engine_flags;
    o.advance_direction = advance_direction;
    return o;
  }
};

// "" => num_runs random sources; "a,b,c" => those; a single value is repeated
// num_runs times (io/parameters.hxx:188-225 of the reference).
inline void parse_source_string(std::string source_str, std::vector<int>* source_vect, int n_vertices, int n_runs) {
  if (source_str.empty()) {
    std::random_device seed;
    std::mt19937 engine(seed());
    std::uniform_int_distribution<int> pick(0, n_vertices - 1);
    for (int i = 0; i < n_runs; ++i) source_vect->push_back(pick(engine));
    return;
  }
  std::stringstream ss(source_str);
  std::string item;
  while (std::getline(ss, item, ',')) {
    int v = -1;
    try {
      v = std::stoi(item);
    } catch (...) {
      v = -1;
    }
    if (v < 0 || v >= n_vertices) {
      std::cout << "Error: Invalid source\n";
      std::exit(1);
    }
    source_vect->push_back(v);
  }
  if (source_vect->size() == 1 && n_runs > 1) source_vect->insert(source_vect->end(), (size_t)n_runs - 1, source_vect->at(0));
}

inline void parse_tag_string(std::string tag_str, std::vector<std::string>* tag_vect) {
  std::stringstream ss(tag_str);
  std::string tag;
  while (std::getline(ss, tag, ','))
    if (!tag.empty()) tag_vect->push_back(tag);
}

}  // namespace cli
}  // namespace io
}  // namespace gunrock
