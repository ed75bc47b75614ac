// fora_neo4j.hpp — host-side mirror of the reference's operator interface for the hot path.
//
// The reference is Java (package joezie.fora_neo4j) and this image has no JDK, so the host side
// above the C ABI is written in C++ with the reference's own class names, method names, argument
// meaning and error behaviour; every method is a thin wrapper over include/pprhip.h (the symbols
// a JNI binding would call — INTEGRATION.md shows that stub).  Cites are relative to
// /root/reference/src/main/java/joezie/fora_neo4j/.
//
// What differs from the Java objects, on purpose:
//   * `Object param` is a double (epsilon / rmax / threshold) — the Integer variants belong to
//     Neo4j_Method, which is outside the hot path;
//   * results are fetched from HBM on demand (getWholeGraphPPR / getTopKNodeIds), the compute
//     calls leave them on the device;
//   * random choices take a seed (the reference uses an unseeded ThreadLocalRandom).
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/pprhip.h"

namespace fora_neo4j {

struct PprError : std::runtime_error {
  int code;
  PprError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// org.neo4j.graphalgo.api.Graph as PPR.setupAdjMatrix builds it (PPR.java:136-152): the one-shot
// lift of the whole graph, here into HBM.  Sources: neo4j-admin-import CSVs or a seeded R-MAT.
class Graph {
 public:
  static std::shared_ptr<Graph> fromNeo4jCsv(const std::string& nodes_csv, const std::string& rels_csv, int device = 0);
  // a Neo4j 3.x store directory such as target/got.db, read without a JVM (PPR.createDb, PPR.java:52-60)
  static std::shared_ptr<Graph> fromNeo4jStore(const std::string& store_dir, int device = 0);
  static std::shared_ptr<Graph> fromRmat(int scale, int edge_factor, uint64_t seed, int device = 0);
  ~Graph();
  long nodeCount() const { return n_; }          // PPR.java:129
  long relationshipCount() const { return m_; }  // PPR.java:131
  int degreeOut(long v) const { return (int)(out_rp_[v + 1] - out_rp_[v]); }
  std::string nodeName(long v) const;  // Algo_Util.getNodeName (Algo_Util.java:21-30)
  pprhip_graph_t* handle() const { return g_; }
  double loadMillis() const { return load_ms_; }

 private:
  Graph() = default;
  void lift(int device, const std::vector<int32_t>& src, const std::vector<int32_t>& dst, bool newest_first);
  void liftEdgelist(int device, pprhip_edgelist_t* el);
  pprhip_graph_t* g_ = nullptr;
  uint32_t n_ = 0;
  uint64_t m_ = 0;
  std::vector<uint32_t> out_rp_;
  std::vector<std::string> names_;
  double load_ms_ = 0.0;
};

using PprMap = std::unordered_map<long, double>;

// Whole_Graph_Util_Interface.java:5-12
struct Whole_Graph_Util_Interface {
  virtual ~Whole_Graph_Util_Interface() = default;
  virtual void computeWholeGraphPPR(long nodeId_start, double param) = 0;
  virtual const PprMap& getWholeGraphPPR() = 0;  // valid until the next compute call
  virtual void printWholeGraphResult() = 0;
};

// Topk_Util_Interface.java:5-15
struct Topk_Util_Interface {
  virtual ~Topk_Util_Interface() = default;
  virtual void computeTopKPPR(long nodeId_start, int k, double param) = 0;
  virtual const std::vector<long>& getTopKNodeIds(int k) = 0;  // sorted by ppr, may exceed k on ties
  virtual void printTopKResult(int k) = 0;
};

// Preprocessing_Interface.java:3-16
struct Preprocessing_Interface {
  virtual ~Preprocessing_Interface() = default;
  virtual void preprocessing(double threshold, double param) = 0;
  virtual void readPreprocessedPPR(long nodeId_start) = 0;
  virtual long getPrepSize() = 0;
  virtual void deletePrepDir() = 0;
};

// Common state of the algorithm objects (Algo_Util.java): graph, alpha, last dense result.
class Algo_Util {
 protected:
  Algo_Util(std::shared_ptr<Graph> adjM, double alpha, std::string dir_db)
      : adjM(std::move(adjM)), alpha(alpha), dir_db(std::move(dir_db)) {}
  void fetchReserve();                       // HBM -> `dense`, `ppr` (entries > 0 only)
  void retrieveTopK(int k, bool from_device);  // Fora_Topk.java:186-199 / Forward_Push.java:413-429
  void printSorted(const char* title, int limit);
  std::shared_ptr<Graph> adjM;
  double alpha;
  std::string dir_db;
  std::vector<double> dense;
  PprMap ppr;
  bool fetched = false;
  std::vector<long> topk_nodeIds;
  PprMap topk_res;
  pprhip_stats_t stats{};

 public:
  const pprhip_stats_t& lastStats() const { return stats; }
};

// Forward_Push.java
class Forward_Push : public Algo_Util, public Whole_Graph_Util_Interface, public Topk_Util_Interface,
                     public Preprocessing_Interface {
 public:
  Forward_Push(double alpha, double rsum, std::shared_ptr<Graph> adjM, std::string dir_db);
  void computeWholeGraphPPR(long nodeId_start, double rmax) override;  // :63-142
  // :144-250; Q lives on the device between rounds (the parked set); isFirstFwdpush = first call after reset
  void forward_push_topk(long nodeId_start, double min_rmax, bool isFirstFwdpush, double rmax);
  double getUpdatedRsum() const { return rsum; }                       // :252-254
  const PprMap& getWholeGraphPPR() override;
  void printWholeGraphResult() override;
  void computeTopKPPR(long nodeId_start, int k, double rmax) override;  // :364-370
  const std::vector<long>& getTopKNodeIds(int k) override;
  void printTopKResult(int k) override;
  void preprocessing(double dummy, double rmax) override;  // :289-340
  void readPreprocessedPPR(long nodeId_start) override;    // :343-362
  long getPrepSize() override;
  void deletePrepDir() override;

 private:
  double rsum;
  std::string preprocessing_dirName;
};

// Monte_Carlo.java
class Monte_Carlo : public Algo_Util, public Whole_Graph_Util_Interface, public Topk_Util_Interface,
                    public Preprocessing_Interface {
 public:
  Monte_Carlo(double alpha, double pfail, double delta, std::shared_ptr<Graph> adjM, std::string dir_db,
              uint64_t seed = 1);
  long random_walk(long nodeId_start);              // :60-94
  long random_walk_no_zero_hop(long nodeId_start);  // :96-133
  void computeWholeGraphPPR(long nodeId_start, double epsilon) override;  // :136-158
  const PprMap& getWholeGraphPPR() override;
  void printWholeGraphResult() override;
  void computeTopKPPR(long nodeId_start, int k, double epsilon) override;
  const std::vector<long>& getTopKNodeIds(int k) override;
  void printTopKResult(int k) override;
  void preprocessing(double dummy, double epsilon) override;  // :181-229
  void readPreprocessedPPR(long nodeId_start) override;       // :232-255
  long getPrepSize() override;
  void deletePrepDir() override;

 private:
  double pfail, delta;
  uint64_t seed, walk_counter = 0;
  std::string preprocessing_dirName;
};

// Fora_Whole_Graph.java
class Fora_Whole_Graph : public Algo_Util, public Whole_Graph_Util_Interface, public Preprocessing_Interface {
 public:
  Fora_Whole_Graph(double alpha, double rsum, double pfail, double delta, std::shared_ptr<Graph> adjM,
                   std::string dir_db, uint64_t seed = 1);
  void computeWholeGraphPPR(long nodeId_start, double epsilon) override;  // :82-146
  const PprMap& getWholeGraphPPR() override;
  void printWholeGraphResult() override;
  void preprocessing(double dummy, double epsilon) override;  // :149-199
  void readPreprocessedPPR(long nodeId_start) override;
  long getPrepSize() override;
  void deletePrepDir() override;
  int rounds = 0;  // 0: the engine's cost model picks the number of threshold halvings (:93-103)

  // Extension over the reference: the harness's query loop (Gen_Util.java:208-232) as one call, 16 queries in
  // flight on the GPU (pprhip_fora_batch_single_source).  selectBatchResult(i) makes result i the current one
  // for getWholeGraphPPR / printWholeGraphResult.
  void computeWholeGraphPPRBatch(const std::vector<long>& sources, double epsilon);
  void selectBatchResult(size_t i);

 private:
  pprhip_fora_conf_t conf{};
  uint64_t seed, query_counter = 0;
  std::string preprocessing_dirName;
  std::vector<double> batch_dense;  // results of the last batch, query-major
};

// Fora_Topk.java
class Fora_Topk : public Algo_Util, public Topk_Util_Interface {
 public:
  Fora_Topk(double alpha, double rsum, double pfail, double delta, double min_delta, int k,
            std::shared_ptr<Graph> adjM, std::string dir_db, uint64_t seed = 1);
  void computeTopKPPR(long nodeId_start, int dummy, double eps) override;  // :102-184 (uses the ctor's k)
  const std::vector<long>& getTopKNodeIds(int dummy) override;             // :82-99
  void printTopKResult(int dummy) override;
  const PprMap& getWholeGraphPPR();

 private:
  pprhip_fora_conf_t conf{};
  int k;
  uint64_t seed, query_counter = 0;
};

// Backward_Search.java
class Backward_Search : public Algo_Util {
 public:
  Backward_Search(double alpha, double rmax, std::shared_ptr<Graph> adjM);
  void backward_search_whole_graph(long nodeId_target);  // :38-100
  const PprMap& getReserve();                            // :102-104

 private:
  double rmax;
};

// Base_Whole_Graph.java
class Base_Whole_Graph : public Algo_Util, public Whole_Graph_Util_Interface, public Topk_Util_Interface,
                         public Preprocessing_Interface {
 public:
  Base_Whole_Graph(double alpha, std::shared_ptr<Graph> adjM, std::string dir_db);
  ~Base_Whole_Graph() override;
  void preprocessing(double threshold, double k) override;                // :58-164 (k < 0: whole graph)
  void computeWholeGraphPPR(long nodeId_start, double dummy) override;    // :167-186 reads <dir>/<id>.txt
  const PprMap& getWholeGraphPPR() override;
  void printWholeGraphResult() override;
  void readPreprocessedPPR(long nodeId_start) override { computeWholeGraphPPR(nodeId_start, 0.0); }
  void computeTopKPPR(long nodeId_start, int, double) override { computeWholeGraphPPR(nodeId_start, 0.0); }
  const std::vector<long>& getTopKNodeIds(int k) override;  // :207-210 file order
  void printTopKResult(int k) override;
  long getPrepSize() override;
  void deletePrepDir() override;

 private:
  std::string preprocessing_dirName;
  std::vector<long> file_order;
};

// Power_Method.java
class Power_Method : public Algo_Util, public Whole_Graph_Util_Interface, public Topk_Util_Interface {
 public:
  Power_Method(double alpha, std::shared_ptr<Graph> adjM, std::string dir_db);
  void computeWholeGraphPPR(long nodeId_start, double dummy) override;  // :44-101, 100 sweeps
  const PprMap& getWholeGraphPPR() override;
  void printWholeGraphResult() override;
  void computeTopKPPR(long nodeId_start, int k, double dummy) override;  // :145-165
  const std::vector<long>& getTopKNodeIds(int k) override;
  const PprMap& getTopK(int) { return topk_res; }                         // :124-126
  void printTopKResult(int k) override;
};

// Algo_Conf.java:25-81
class Algo_Conf {
 public:
  Algo_Conf(double alpha, uint64_t seed = 1) : alpha(alpha), seed(seed) {}
  std::unique_ptr<Power_Method> set_conf_power_method(std::shared_ptr<Graph> adjM, const std::string& dir_db);
  std::unique_ptr<Monte_Carlo> set_conf_mc(std::shared_ptr<Graph> adjM, const std::string& dir_db);
  std::unique_ptr<Base_Whole_Graph> set_conf_base_whole_graph(std::shared_ptr<Graph> adjM, const std::string& dir_db);
  std::unique_ptr<Fora_Whole_Graph> set_conf_fora_whole_graph(std::shared_ptr<Graph> adjM, const std::string& dir_db);
  std::unique_ptr<Forward_Push> set_conf_fwdpush(std::shared_ptr<Graph> adjM, const std::string& dir_db);
  std::unique_ptr<Fora_Topk> set_conf_fora_topk(int k, std::shared_ptr<Graph> adjM, const std::string& dir_db);
  double alpha, delta = 0, pfail = 0, rsum = 0, min_delta = 0;
  int k = 0;
  uint64_t seed;
};

// Gen_Util.java: query sampling, error metrics, the timing loops and the report writer.
enum class AlgoType { POWER_METHOD, FORA_WHOLE_GRAPH, FORA_TOPK, FWDPUSH, MC, BASE_WHOLE_GRAPH };
enum class TestType { WHOLE_GRAPH, TOPK };
const char* algoName(AlgoType t);

class Gen_Util {
 public:
  Gen_Util(std::shared_ptr<Graph> adjM, double alpha, std::string dir_db, uint64_t seed);
  std::vector<long> getQueryNodes(int query_num);  // :99-107, seeded
  // :259-326
  static double maxErr(const PprMap& algo, const PprMap& gnd);
  static double precision(const std::vector<long>& algo, const std::vector<long>& gnd);
  static double ndcg(const std::vector<long>& algo, const std::vector<long>& gnd, const PprMap& gnd_topk);
  // :109-257; appends one report row to <db>_AlgoPerfResults.txt in the reference's column order
  void algo_perf_test(AlgoType algoType, int query_num, int k, double param, double threshold, bool to_be_preprocessed,
                      TestType testType);
  // :328-647 with the Game-of-Thrones parameter set (Testset5, :451-478)
  void algo_perf_batch_test(int query_num, int k);
  std::string report_file;

 private:
  std::shared_ptr<Graph> adjM;
  double alpha;
  std::string dir_db;
  uint64_t seed, draws = 0;
};

}  // namespace fora_neo4j
