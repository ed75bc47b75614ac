// ppr_main.cpp — native counterpart of PPR.main (PPR.java:154-200): same flags, same defaults,
// same batch of experiments and the same report file, with the graph lifted into HBM instead of
// HeavyGraph and every algorithm running on the GPU through the C ABI.
//
//   ppr -alpha 0.15 -eps 0.5 -query 50 -k 10 -db <dir>
//
// -db names a Neo4j 3.x store directory (e.g. target/got.db: the node and relationship record
// files are read directly, no JVM), a directory holding neo4j-admin-import CSVs
// (<X>_Nodes.csv / <X>_Rels.csv, e.g. dataset/got) or "rmat:<scale>[:<seed>]".
// -node/-label/-rel are accepted for command-line compatibility; the lift takes every label and
// relationship type, as PPR.setupAdjMatrix does (PPR.java:141-147).
#include <dirent.h>

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "fora_neo4j.hpp"

using namespace fora_neo4j;

static void help() {
  std::cout << "usage: PPR\n"
               " -alpha <arg>   The possibility that a random walk stops at current node (Default: 0.15)\n"
               " -db <arg>      The directory of the input database (Default: \"target/got.db\")\n"
               " -eps <arg>     The relative error bound (Default: 0.5)\n"
               " -help          Print information about command line inputs.\n"
               " -k <arg>       For Top-k Algorithm tests: the number of nodes with greatest PPR value that\n"
               "                we're interested in (Default: 10)\n"
               " -label <arg>   The nodes' label type in the input datatbase (Default: \"Person\")\n"
               " -node <arg>    The node property in the input database (Default: \"name\")\n"
               " -query <arg>   The number of queries for the test (Default: 50)\n"
               " -rel <arg>     The relationships' type in the input database (Default: \"Relation\")\n"
               " -seed <arg>    Seed of the query sampler and the walks (Default: 1)\n"
               " -single <id>   Print FORA whole-graph and top-k results of one source instead of the batch\n";
}

static bool endsWith(const std::string& s, const std::string& t) {
  return s.size() >= t.size() && s.compare(s.size() - t.size(), t.size(), t) == 0;
}

int main(int argc, char** argv) {
  // Kernel arguments in device memory (+4-10 % on the chains of short kernels).  The HIP runtime
  // reads the switch when it initialises; here no thread and no HIP call exists yet, and a value the user has
  // exported wins.
  (void)setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
  double alpha = 0.15, eps = 0.5;
  int query = 50, k = 10;
  long single = -1;
  uint64_t seed = 1;
  std::string node = "name", label = "Person", rel = "Relation", db = "target/got.db";
  try {
    for (int i = 1; i < argc; ++i) {
      std::string a = argv[i];
      while (!a.empty() && a[0] == '-') a.erase(0, 1);
      if (a == "help") {
        help();
        return 0;
      }
      if (i + 1 >= argc) throw PprError(PPRHIP_ERR_INVALID, "Missing argument for option: " + a);
      std::string v = argv[++i];
      if (a == "alpha") alpha = std::stod(v);
      else if (a == "eps") eps = std::stod(v);
      else if (a == "query") query = std::stoi(v);
      else if (a == "k") k = std::stoi(v);
      else if (a == "node") node = v;
      else if (a == "label") label = v;
      else if (a == "rel") rel = v;
      else if (a == "db") db = v;
      else if (a == "seed") seed = std::stoull(v);
      else if (a == "single") single = std::stol(v);
      else throw PprError(PPRHIP_ERR_INVALID, "Unrecognized option: -" + a);
    }
    std::string dir_db = db;  // createDb keeps the directory's base name (PPR.java:52-60)
    while (dir_db.size() > 1 && dir_db.back() == '/') dir_db.pop_back();
    size_t slash = dir_db.find_last_of('/');
    if (slash != std::string::npos) dir_db = dir_db.substr(slash + 1);

    std::cout << "\nLoading graph from database..." << std::endl;
    std::shared_ptr<Graph> adjM;
    if (db.rfind("rmat:", 0) == 0) {
      int scale = std::atoi(db.c_str() + 5);
      size_t c2 = db.find(':', 5);
      uint64_t gseed = c2 == std::string::npos ? 1 : std::stoull(db.substr(c2 + 1));
      adjM = Graph::fromRmat(scale, 16, gseed);
      dir_db = "rmat" + std::to_string(scale);
    } else {
      std::string nodes, rels;
      bool store = false;
      if (FILE* f = fopen((db + "/neostore.nodestore.db").c_str(), "rb")) {
        fclose(f);
        store = true;
      }
      if (store) {
        adjM = Graph::fromNeo4jStore(db);
      } else if (DIR* d = opendir(db.c_str())) {
        while (dirent* e = readdir(d)) {
          std::string f = e->d_name;
          if (endsWith(f, "_Nodes.csv")) nodes = db + "/" + f;
          if (endsWith(f, "_Rels.csv")) rels = db + "/" + f;
        }
        closedir(d);
      }
      if (!store) {
        if (nodes.empty() || rels.empty())
          throw PprError(PPRHIP_ERR_IO, "neither a Neo4j store nor *_Nodes.csv / *_Rels.csv under " + db);
        adjM = Graph::fromNeo4jCsv(nodes, rels);
      }
    }
    std::cout << "\nFinish graph loading in " << (long)adjM->loadMillis() << "(ms)" << std::endl;
    std::cout << "node_amount = " << adjM->nodeCount() << ", rel_amount = " << adjM->relationshipCount() << std::endl;

    if (single >= 0) {
      Algo_Conf conf(alpha, seed);
      auto fora = conf.set_conf_fora_whole_graph(adjM, dir_db);
      fora->computeWholeGraphPPR(single, eps);
      fora->printWholeGraphResult();
      auto topk = conf.set_conf_fora_topk(k, adjM, dir_db);
      topk->computeTopKPPR(single, k, eps);
      topk->printTopKResult(k);
      return 0;
    }
    Gen_Util gen(adjM, alpha, dir_db, seed);
    gen.algo_perf_batch_test(query, k);
    std::cout << "\nResults appended to " << gen.report_file << std::endl;
  } catch (const std::exception& e) {  // PPR.java:196-199: report and end normally
    std::cout << "Algo performance batch test failed!" << std::endl;
    std::cout << e.what() << std::endl;
  }
  return 0;
}
