// fora_neo4j.cpp — implementation of the host-side mirror (see fora_neo4j.hpp); every compute
// method is one call into the C ABI of include/pprhip.h.
#include "fora_neo4j.hpp"

#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <sstream>

namespace fora_neo4j {

static void check(int rc) {
  if (rc != PPRHIP_OK) throw PprError(rc, std::string("pprhip: ") + pprhip_last_error());
}

static std::string jdouble(double d) {  // Double.toString
  char buf[64];
  if (pprhip_format_double(d, buf, sizeof buf) < 0) return "NaN";
  return buf;
}

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static uint64_t splitmix(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// ------------------------------------------------------------------ directories (commons-io in the reference)
static void mkdirs(const std::string& path) {
  for (size_t i = 1; i <= path.size(); ++i)
    if (i == path.size() || path[i] == '/') mkdir(path.substr(0, i).c_str(), 0777);
}
static long dirSize(const std::string& path) {  // FileUtils.sizeOfDirectory
  long total = 0;
  DIR* d = opendir(path.c_str());
  if (!d) return 0;
  while (dirent* e = readdir(d)) {
    if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
    std::string p = path + "/" + e->d_name;
    struct stat st;
    if (stat(p.c_str(), &st) != 0) continue;
    total += S_ISDIR(st.st_mode) ? dirSize(p) : (long)st.st_size;
  }
  closedir(d);
  return total;
}
static void rmTree(const std::string& path, bool keep_root) {  // FileUtils.cleanDirectory / deleteDirectory
  DIR* d = opendir(path.c_str());
  if (!d) return;
  while (dirent* e = readdir(d)) {
    if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
    std::string p = path + "/" + e->d_name;
    struct stat st;
    if (stat(p.c_str(), &st) != 0) continue;
    if (S_ISDIR(st.st_mode)) rmTree(p, false); else unlink(p.c_str());
  }
  closedir(d);
  if (!keep_root) rmdir(path.c_str());
}
static void writeMapFile(const std::string& file, const std::vector<std::pair<long, double>>& rows) {
  std::ofstream f(file, std::ios::trunc);
  for (auto& r : rows) f << r.first << '\t' << jdouble(r.second) << '\n';  // "<id>\t<Double.toString>\n"
}
static void readMapFile(const std::string& file, PprMap& out, std::vector<long>* order) {
  out.clear();
  if (order) order->clear();
  std::ifstream f(file);
  if (!f) {
    std::cout << "Read from file " << file << " failed!" << std::endl;  // the reference prints and carries on
    return;
  }
  std::string line;
  while (std::getline(f, line)) {
    size_t tab = line.find('\t');
    if (tab == std::string::npos) continue;
    long id = std::stol(line.substr(0, tab));
    out[id] = std::stod(line.substr(tab + 1));
    if (order) order->push_back(id);
  }
}

// ------------------------------------------------------------------ Graph
void Graph::lift(int device, const std::vector<int32_t>& src, const std::vector<int32_t>& dst, bool newest_first) {
  const double t0 = now_ms();
  std::vector<uint32_t> in_rp((size_t)n_ + 1);
  std::vector<int32_t> out_ci(std::max<size_t>(1, m_)), in_ci(std::max<size_t>(1, m_));
  out_rp_.resize((size_t)n_ + 1);
  check(pprhip_csr_build(n_, m_, src.data(), dst.data(), newest_first, out_rp_.data(), out_ci.data()));
  check(pprhip_csr_build(n_, m_, dst.data(), src.data(), newest_first, in_rp.data(), in_ci.data()));
  check(pprhip_graph_create(n_, m_, out_rp_.data(), out_ci.data(), in_rp.data(), in_ci.data(), device, &g_));
  load_ms_ = now_ms() - t0;
}

std::shared_ptr<Graph> Graph::fromNeo4jCsv(const std::string& nodes_csv, const std::string& rels_csv, int device) {
  std::shared_ptr<Graph> g(new Graph());
  pprhip_edgelist_t* el = nullptr;
  check(pprhip_edgelist_from_neo4j_csv(nodes_csv.c_str(), rels_csv.c_str(), &el));
  const int32_t *s = nullptr, *d = nullptr;
  check(pprhip_edgelist_info(el, &g->n_, &g->m_));
  check(pprhip_edgelist_edges(el, &s, &d));
  std::vector<int32_t> src(s, s + g->m_), dst(d, d + g->m_);
  for (uint32_t v = 0; v < g->n_; ++v) g->names_.push_back(pprhip_edgelist_node_name(el, v));
  pprhip_edgelist_destroy(el);
  g->lift(device, src, dst, true);  // HeavyGraph lists the newest relationship first
  return g;
}

void Graph::liftEdgelist(int device, pprhip_edgelist_t* el) {
  const double t0 = now_ms();
  check(pprhip_edgelist_info(el, &n_, &m_));
  std::vector<uint32_t> in_rp((size_t)n_ + 1);
  std::vector<int32_t> out_ci(std::max<size_t>(1, m_)), in_ci(std::max<size_t>(1, m_));
  out_rp_.resize((size_t)n_ + 1);
  check(pprhip_edgelist_build_csr(el, 0, out_rp_.data(), out_ci.data()));
  check(pprhip_edgelist_build_csr(el, 1, in_rp.data(), in_ci.data()));
  for (uint32_t v = 0; v < n_; ++v) names_.push_back(pprhip_edgelist_node_name(el, v));
  check(pprhip_graph_create(n_, m_, out_rp_.data(), out_ci.data(), in_rp.data(), in_ci.data(), device, &g_));
  load_ms_ = now_ms() - t0;
}

std::shared_ptr<Graph> Graph::fromNeo4jStore(const std::string& store_dir, int device) {
  std::shared_ptr<Graph> g(new Graph());
  pprhip_edgelist_t* el = nullptr;
  check(pprhip_edgelist_from_neo4j_store(store_dir.c_str(), &el));
  try {
    g->liftEdgelist(device, el);
  } catch (...) {
    pprhip_edgelist_destroy(el);
    throw;
  }
  pprhip_edgelist_destroy(el);
  return g;
}

std::shared_ptr<Graph> Graph::fromRmat(int scale, int edge_factor, uint64_t seed, int device) {
  std::shared_ptr<Graph> g(new Graph());
  g->n_ = 1u << scale;
  g->m_ = (uint64_t)edge_factor << scale;
  std::vector<int32_t> src(g->m_), dst(g->m_);
  check(pprhip_rmat_edges(scale, edge_factor, seed, src.data(), dst.data()));
  g->lift(device, src, dst, false);
  return g;
}

Graph::~Graph() { pprhip_graph_destroy(g_); }

std::string Graph::nodeName(long v) const {
  if (v >= 0 && (size_t)v < names_.size()) return names_[v];
  return std::to_string(v);
}

// ------------------------------------------------------------------ Algo_Util
void Algo_Util::fetchReserve() {
  if (fetched) return;
  dense.resize(adjM->nodeCount());
  check(pprhip_get_reserve(adjM->handle(), dense.data()));
  ppr.clear();
  for (size_t v = 0; v < dense.size(); ++v)
    if (dense[v] > 0.0) ppr[(long)v] = dense[v];  // the reference's map holds touched nodes only
  fetched = true;
}

void Algo_Util::retrieveTopK(int k, bool from_device) {
  topk_nodeIds.clear();
  topk_res.clear();
  if (from_device) {
    // all entries >= the k-th value (may exceed k), sorted by value; first ask how many there are
    int n_sel = 0;
    double kth = 0.0;
    std::vector<int32_t> ids(std::max(1, k));
    std::vector<double> vals(std::max(1, k));
    check(pprhip_topk_select(adjM->handle(), k, ids.data(), vals.data(), k, &n_sel, &kth, nullptr));
    if (n_sel > k) {
      ids.resize(n_sel);
      vals.resize(n_sel);
      check(pprhip_topk_select(adjM->handle(), k, ids.data(), vals.data(), n_sel, &n_sel, &kth, nullptr));
    }
    for (int i = 0; i < n_sel; ++i) {
      topk_nodeIds.push_back(ids[i]);
      topk_res[ids[i]] = vals[i];
    }
    return;
  }
  std::vector<std::pair<long, double>> rows(ppr.begin(), ppr.end());
  std::sort(rows.begin(), rows.end(), [](auto& a, auto& b) { return a.second != b.second ? a.second > b.second : a.first < b.first; });
  double kth = (k >= 1 && (size_t)k <= rows.size()) ? rows[k - 1].second : -1.0;
  for (auto& r : rows)
    if (kth < 0 || r.second >= kth) {
      topk_nodeIds.push_back(r.first);
      topk_res[r.first] = r.second;
    }
}

void Algo_Util::printSorted(const char* title, int limit) {
  std::vector<std::pair<long, double>> rows(ppr.begin(), ppr.end());
  std::sort(rows.begin(), rows.end(), [](auto& a, auto& b) { return a.second != b.second ? a.second > b.second : a.first < b.first; });
  std::cout << title << std::endl;
  int count = 0;
  for (auto& r : rows) {
    if (limit >= 0 && count++ >= limit) break;
    std::cout << "@" << adjM->nodeName(r.first) << '\t' << jdouble(r.second) << std::endl;
  }
}

// ------------------------------------------------------------------ Forward_Push
Forward_Push::Forward_Push(double alpha, double rsum, std::shared_ptr<Graph> adjM, std::string dir_db)
    : Algo_Util(std::move(adjM), alpha, dir_db), rsum(rsum), preprocessing_dirName("FWP_ppr_results/" + dir_db) {}

void Forward_Push::computeWholeGraphPPR(long s, double rmax) {
  fetched = false;
  topk_nodeIds.clear();
  topk_res.clear();
  check(pprhip_forward_push(adjM->handle(), (int32_t)s, alpha, rmax, nullptr, nullptr, &rsum, &stats));
}

void Forward_Push::forward_push_topk(long s, double min_rmax, bool isFirstFwdpush, double rmax) {
  fetched = false;
  if (isFirstFwdpush) check(pprhip_fwdpush_topk_reset(adjM->handle(), (int32_t)s, alpha));
  check(pprhip_fwdpush_topk_round(adjM->handle(), min_rmax, rmax, &rsum, &stats));
}

const PprMap& Forward_Push::getWholeGraphPPR() {
  fetchReserve();
  return ppr;
}
void Forward_Push::printWholeGraphResult() {
  fetchReserve();
  printSorted("Forward-Push Reserve(pi):", -1);
}
void Forward_Push::computeTopKPPR(long s, int, double rmax) { computeWholeGraphPPR(s, rmax); }
const std::vector<long>& Forward_Push::getTopKNodeIds(int k) {
  if (topk_nodeIds.empty()) retrieveTopK(k, true);
  return topk_nodeIds;
}
void Forward_Push::printTopKResult(int k) {
  getTopKNodeIds(k);
  std::cout << "\nForward-Push-Top" << k << " PPR:" << std::endl;
  for (int i = 0; i < k && i < (int)topk_nodeIds.size(); ++i)
    std::cout << "@" << adjM->nodeName(topk_nodeIds[i]) << '\t' << jdouble(topk_res[topk_nodeIds[i]]) << std::endl;
}
void Forward_Push::preprocessing(double, double rmax) {
  preprocessing_dirName += "/" + jdouble(rmax);
  mkdirs(preprocessing_dirName);
  rmTree(preprocessing_dirName, true);
  for (long v = 0; v < adjM->nodeCount(); ++v) {
    computeWholeGraphPPR(v, rmax);
    fetchReserve();
    std::vector<std::pair<long, double>> rows(ppr.begin(), ppr.end());
    std::sort(rows.begin(), rows.end());
    writeMapFile(preprocessing_dirName + "/" + std::to_string(v) + ".txt", rows);
  }
}
void Forward_Push::readPreprocessedPPR(long s) {
  readMapFile(preprocessing_dirName + "/" + std::to_string(s) + ".txt", ppr, nullptr);
  fetched = true;
}
long Forward_Push::getPrepSize() { return dirSize(preprocessing_dirName); }
void Forward_Push::deletePrepDir() { rmTree(preprocessing_dirName, false); }

// ------------------------------------------------------------------ Monte_Carlo
Monte_Carlo::Monte_Carlo(double alpha, double pfail, double delta, std::shared_ptr<Graph> adjM, std::string dir_db,
                         uint64_t seed)
    : Algo_Util(std::move(adjM), alpha, dir_db), pfail(pfail), delta(delta), seed(seed),
      preprocessing_dirName("MC_ppr_results/" + dir_db) {}  // Monte_Carlo.java:57

long Monte_Carlo::random_walk(long s) {
  int32_t start = (int32_t)s, term = 0;
  uint64_t idx = walk_counter++;
  check(pprhip_random_walk_batch(adjM->handle(), &start, &idx, 1, alpha, seed, 0, 0, &term, nullptr));
  return term;
}
long Monte_Carlo::random_walk_no_zero_hop(long s) {
  int32_t start = (int32_t)s, term = 0;
  uint64_t idx = walk_counter++;
  check(pprhip_random_walk_batch(adjM->handle(), &start, &idx, 1, alpha, seed, 0, 1, &term, nullptr));
  return term;
}
void Monte_Carlo::computeWholeGraphPPR(long s, double epsilon) {
  fetched = false;
  topk_nodeIds.clear();
  topk_res.clear();
  pprhip_fora_conf_t c{};
  c.alpha = alpha;
  c.delta = delta;
  c.pfail = pfail;
  c.rsum = 1.0;
  c.n = (uint32_t)adjM->nodeCount();
  c.m = (uint64_t)adjM->relationshipCount();
  check(pprhip_monte_carlo(adjM->handle(), (int32_t)s, epsilon, &c, seed + (walk_counter++), nullptr, &stats));
}
const PprMap& Monte_Carlo::getWholeGraphPPR() {
  fetchReserve();
  return ppr;
}
void Monte_Carlo::printWholeGraphResult() {
  fetchReserve();
  printSorted("Monte-Carlo PPR:", -1);
}
void Monte_Carlo::computeTopKPPR(long s, int, double epsilon) { computeWholeGraphPPR(s, epsilon); }
const std::vector<long>& Monte_Carlo::getTopKNodeIds(int k) {
  if (topk_nodeIds.empty()) retrieveTopK(k, true);
  return topk_nodeIds;
}
void Monte_Carlo::printTopKResult(int k) {
  getTopKNodeIds(k);
  std::cout << "\nMonte-Carlo-Top" << k << " PPR:" << std::endl;
  for (int i = 0; i < k && i < (int)topk_nodeIds.size(); ++i)
    std::cout << "@" << adjM->nodeName(topk_nodeIds[i]) << '\t' << jdouble(topk_res[topk_nodeIds[i]]) << std::endl;
}

void Monte_Carlo::preprocessing(double, double epsilon) {  // Monte_Carlo.java:181-229
  preprocessing_dirName += "/" + jdouble(epsilon);
  std::cout << "\nMonte-Carlo preprocessing started..." << std::endl;
  mkdirs(preprocessing_dirName);
  rmTree(preprocessing_dirName, true);
  for (long v = 0; v < adjM->nodeCount(); ++v) {  // adjM.forEachNode (:209)
    computeWholeGraphPPR(v, epsilon);
    fetchReserve();
    std::vector<std::pair<long, double>> rows(ppr.begin(), ppr.end());
    std::sort(rows.begin(), rows.end());  // the reference writes HashMap order; ascending ids here
    writeMapFile(preprocessing_dirName + "/" + std::to_string(v) + ".txt", rows);
  }
}
void Monte_Carlo::readPreprocessedPPR(long s) {  // :232-255
  readMapFile(preprocessing_dirName + "/" + std::to_string(s) + ".txt", ppr, nullptr);
  fetched = true;
}
long Monte_Carlo::getPrepSize() { return dirSize(preprocessing_dirName); }
void Monte_Carlo::deletePrepDir() { rmTree(preprocessing_dirName, false); }

// ------------------------------------------------------------------ Fora_Whole_Graph
Fora_Whole_Graph::Fora_Whole_Graph(double alpha, double rsum, double pfail, double delta, std::shared_ptr<Graph> adjM,
                                   std::string dir_db, uint64_t seed)
    : Algo_Util(std::move(adjM), alpha, dir_db), seed(seed), preprocessing_dirName("FORA_ppr_results/" + dir_db) {
  conf.alpha = alpha;
  conf.delta = delta;
  conf.pfail = pfail;
  conf.rsum = rsum;
  conf.n = (uint32_t)this->adjM->nodeCount();
  conf.m = (uint64_t)this->adjM->relationshipCount();
}
void Fora_Whole_Graph::computeWholeGraphPPR(long s, double epsilon) {
  fetched = false;
  check(pprhip_fora_single_source(adjM->handle(), (int32_t)s, epsilon, &conf, seed + (query_counter++), rounds,
                                  nullptr, &stats));
}
void Fora_Whole_Graph::computeWholeGraphPPRBatch(const std::vector<long>& sources, double epsilon) {
  fetched = false;
  const size_t n = (size_t)adjM->nodeCount();
  std::vector<int32_t> srcs(sources.begin(), sources.end());
  batch_dense.assign(sources.size() * n, 0.0);
  pprhip_tuning_t saved, batch;
  check(pprhip_graph_get_tuning(adjM->handle(), &saved));
  pprhip_tuning_batch(&batch);  // a dense level costs a query 1/16 of a sweep
  check(pprhip_graph_set_tuning(adjM->handle(), &batch));
  const int rc = pprhip_fora_batch_single_source(adjM->handle(), srcs.data(), (int)srcs.size(), epsilon, &conf,
                                                 seed + query_counter, rounds, batch_dense.data(), 0, nullptr, nullptr,
                                                 nullptr, nullptr, &stats);
  (void)pprhip_graph_set_tuning(adjM->handle(), &saved);
  check(rc);
  query_counter += sources.size();
}
void Fora_Whole_Graph::selectBatchResult(size_t i) {
  const size_t n = (size_t)adjM->nodeCount();
  if ((i + 1) * n > batch_dense.size()) throw PprError(PPRHIP_ERR_INVALID, "selectBatchResult: no such batch result");
  dense.assign(batch_dense.begin() + i * n, batch_dense.begin() + (i + 1) * n);
  ppr.clear();
  for (size_t v = 0; v < n; ++v)
    if (dense[v] > 0.0) ppr[(long)v] = dense[v];
  fetched = true;
}
const PprMap& Fora_Whole_Graph::getWholeGraphPPR() {
  fetchReserve();
  return ppr;
}
void Fora_Whole_Graph::printWholeGraphResult() {
  fetchReserve();
  printSorted("Fora-Whole-Graph PPR:", -1);
}
void Fora_Whole_Graph::preprocessing(double, double epsilon) {
  preprocessing_dirName += "/" + jdouble(epsilon);
  mkdirs(preprocessing_dirName);
  rmTree(preprocessing_dirName, true);
  // every node is a source (:160-197): 64 at a time through the batched entry point
  const long n = adjM->nodeCount();
  for (long v0 = 0; v0 < n; v0 += 64) {
    std::vector<long> chunk;
    for (long v = v0; v < std::min(n, v0 + 64); ++v) chunk.push_back(v);
    computeWholeGraphPPRBatch(chunk, epsilon);
    for (size_t i = 0; i < chunk.size(); ++i) {
      selectBatchResult(i);
      std::vector<std::pair<long, double>> rows(ppr.begin(), ppr.end());
      std::sort(rows.begin(), rows.end());
      writeMapFile(preprocessing_dirName + "/" + std::to_string(chunk[i]) + ".txt", rows);
    }
  }
}
void Fora_Whole_Graph::readPreprocessedPPR(long s) {
  readMapFile(preprocessing_dirName + "/" + std::to_string(s) + ".txt", ppr, nullptr);
  fetched = true;
}
long Fora_Whole_Graph::getPrepSize() { return dirSize(preprocessing_dirName); }
void Fora_Whole_Graph::deletePrepDir() { rmTree(preprocessing_dirName, false); }

// ------------------------------------------------------------------ Fora_Topk
Fora_Topk::Fora_Topk(double alpha, double rsum, double pfail, double delta, double min_delta, int k,
                     std::shared_ptr<Graph> adjM, std::string dir_db, uint64_t seed)
    : Algo_Util(std::move(adjM), alpha, dir_db), k(k), seed(seed) {
  conf.alpha = alpha;
  conf.delta = delta;
  conf.pfail = pfail;
  conf.rsum = rsum;
  conf.min_delta = min_delta;
  conf.k = k;
  conf.n = (uint32_t)this->adjM->nodeCount();
  conf.m = (uint64_t)this->adjM->relationshipCount();
}
void Fora_Topk::computeTopKPPR(long s, int, double eps) {
  fetched = false;
  topk_nodeIds.clear();
  topk_res.clear();
  int n_sel = 0;
  std::vector<int32_t> ids(k);
  std::vector<double> vals(k);
  check(pprhip_fora_topk(adjM->handle(), (int32_t)s, eps, &conf, seed + (query_counter++), ids.data(), vals.data(), k,
                         &n_sel, nullptr, &stats));
  if (n_sel <= k) {  // no ties beyond k: the answer is already here
    for (int i = 0; i < n_sel; ++i) {
      topk_nodeIds.push_back(ids[i]);
      topk_res[ids[i]] = vals[i];
    }
    if (n_sel == 0) topk_nodeIds.clear();
  }
}
const std::vector<long>& Fora_Topk::getTopKNodeIds(int) {
  if (topk_nodeIds.empty()) retrieveTopK(k, true);
  return topk_nodeIds;
}
void Fora_Topk::printTopKResult(int) {
  getTopKNodeIds(k);
  std::cout << "\nFora-Top" << k << " PPR:" << std::endl;
  for (int i = 0; i < k && i < (int)topk_nodeIds.size(); ++i)
    std::cout << "@" << adjM->nodeName(topk_nodeIds[i]) << '\t' << jdouble(topk_res[topk_nodeIds[i]]) << std::endl;
}
const PprMap& Fora_Topk::getWholeGraphPPR() {
  fetchReserve();
  return ppr;
}

// ------------------------------------------------------------------ Backward_Search
Backward_Search::Backward_Search(double alpha, double rmax, std::shared_ptr<Graph> adjM)
    : Algo_Util(std::move(adjM), alpha, ""), rmax(rmax) {}
void Backward_Search::backward_search_whole_graph(long t) {
  fetched = false;
  check(pprhip_backward_push(adjM->handle(), (int32_t)t, alpha, rmax, nullptr, nullptr, &stats));
}
const PprMap& Backward_Search::getReserve() {
  fetchReserve();
  return ppr;
}

// ------------------------------------------------------------------ Base_Whole_Graph
Base_Whole_Graph::Base_Whole_Graph(double alpha, std::shared_ptr<Graph> adjM, std::string dir_db)
    : Algo_Util(std::move(adjM), alpha, dir_db), preprocessing_dirName("BASE_ppr_results/" + dir_db) {}
Base_Whole_Graph::~Base_Whole_Graph() = default;

void Base_Whole_Graph::preprocessing(double threshold, double kd) {
  const int k = (int)kd;
  preprocessing_dirName += "/" + jdouble(threshold) + "_" + std::to_string(k);  // Base_Whole_Graph.java:65
  std::cout << "\nBASE preprocessing starts..." << std::endl;
  pprhip_index_t* ix = nullptr;
  check(pprhip_all_pair_backward(adjM->handle(), alpha, threshold, k, 0, (uint32_t)adjM->nodeCount(), &ix, &stats));
  mkdirs(preprocessing_dirName);
  rmTree(preprocessing_dirName, true);
  std::cout << "\nBASE preprocessing storing results to files (under directory " << preprocessing_dirName << ")..."
            << std::endl;
  int rc = pprhip_index_write_dir(ix, preprocessing_dirName.c_str());
  pprhip_index_destroy(ix);
  check(rc);
}
void Base_Whole_Graph::computeWholeGraphPPR(long s, double) {
  readMapFile(preprocessing_dirName + "/" + std::to_string(s) + ".txt", ppr, &file_order);
  fetched = true;
}
const PprMap& Base_Whole_Graph::getWholeGraphPPR() { return ppr; }
void Base_Whole_Graph::printWholeGraphResult() {
  std::cout << "\nBase-Whole-Graph PPR:" << std::endl;
  for (long id : file_order) std::cout << "@" << adjM->nodeName(id) << '\t' << jdouble(ppr[id]) << std::endl;
}
const std::vector<long>& Base_Whole_Graph::getTopKNodeIds(int) { return file_order; }
void Base_Whole_Graph::printTopKResult(int k) {
  std::cout << "\nBase-Top" << k << " PPR:" << std::endl;
  for (int i = 0; i < k && i < (int)file_order.size(); ++i)
    std::cout << "@" << adjM->nodeName(file_order[i]) << '\t' << jdouble(ppr[file_order[i]]) << std::endl;
}
long Base_Whole_Graph::getPrepSize() { return dirSize(preprocessing_dirName); }
void Base_Whole_Graph::deletePrepDir() { rmTree(preprocessing_dirName, false); }

// ------------------------------------------------------------------ Power_Method
Power_Method::Power_Method(double alpha, std::shared_ptr<Graph> adjM, std::string dir_db)
    : Algo_Util(std::move(adjM), alpha, dir_db) {}
void Power_Method::computeWholeGraphPPR(long s, double) {
  fetched = false;
  topk_nodeIds.clear();
  topk_res.clear();
  check(pprhip_power_method(adjM->handle(), (int32_t)s, alpha, 100, nullptr, &stats));  // Power_Method.java:55
}
const PprMap& Power_Method::getWholeGraphPPR() {
  fetchReserve();
  return ppr;
}
void Power_Method::printWholeGraphResult() {
  fetchReserve();
  printSorted("Exact-PPR:", -1);
}
void Power_Method::computeTopKPPR(long s, int k, double d) {
  computeWholeGraphPPR(s, d);
  retrieveTopK(k, true);
}
const std::vector<long>& Power_Method::getTopKNodeIds(int) { return topk_nodeIds; }
void Power_Method::printTopKResult(int k) {
  std::cout << "\nExact-Top" << k << " PPR:" << std::endl;
  for (int i = 0; i < k && i < (int)topk_nodeIds.size(); ++i)
    std::cout << "@" << adjM->nodeName(topk_nodeIds[i]) << '\t' << jdouble(topk_res[topk_nodeIds[i]]) << std::endl;
}

// ------------------------------------------------------------------ Algo_Conf (Algo_Conf.java:25-81)
std::unique_ptr<Power_Method> Algo_Conf::set_conf_power_method(std::shared_ptr<Graph> adjM, const std::string& db) {
  return std::make_unique<Power_Method>(alpha, adjM, db);
}
std::unique_ptr<Monte_Carlo> Algo_Conf::set_conf_mc(std::shared_ptr<Graph> adjM, const std::string& db) {
  delta = pfail = 1.0 / (double)adjM->nodeCount();
  rsum = 1.0;
  return std::make_unique<Monte_Carlo>(alpha, pfail, delta, adjM, db, seed);
}
std::unique_ptr<Base_Whole_Graph> Algo_Conf::set_conf_base_whole_graph(std::shared_ptr<Graph> adjM, const std::string& db) {
  delta = pfail = 1.0 / (double)adjM->nodeCount();
  return std::make_unique<Base_Whole_Graph>(alpha, adjM, db);
}
std::unique_ptr<Fora_Whole_Graph> Algo_Conf::set_conf_fora_whole_graph(std::shared_ptr<Graph> adjM, const std::string& db) {
  pprhip_fora_conf_t c;
  check(pprhip_conf_fora_whole_graph((uint32_t)adjM->nodeCount(), (uint64_t)adjM->relationshipCount(), alpha, &c));
  delta = c.delta;
  pfail = c.pfail;
  rsum = c.rsum;
  return std::make_unique<Fora_Whole_Graph>(alpha, rsum, pfail, delta, adjM, db, seed);
}
std::unique_ptr<Forward_Push> Algo_Conf::set_conf_fwdpush(std::shared_ptr<Graph> adjM, const std::string& db) {
  delta = pfail = 1.0 / (double)adjM->nodeCount();
  rsum = 1.0;
  return std::make_unique<Forward_Push>(alpha, rsum, adjM, db);
}
std::unique_ptr<Fora_Topk> Algo_Conf::set_conf_fora_topk(int k_, std::shared_ptr<Graph> adjM, const std::string& db) {
  pprhip_fora_conf_t c;
  check(pprhip_conf_fora_topk((uint32_t)adjM->nodeCount(), (uint64_t)adjM->relationshipCount(), k_, alpha, &c));
  k = k_;
  min_delta = c.min_delta;
  delta = c.delta;
  pfail = c.pfail;
  rsum = c.rsum;
  return std::make_unique<Fora_Topk>(alpha, rsum, pfail, delta, min_delta, k, adjM, db, seed);
}

// ------------------------------------------------------------------ Gen_Util
const char* algoName(AlgoType t) {
  switch (t) {
    case AlgoType::POWER_METHOD: return "POWER_METHOD";
    case AlgoType::FORA_WHOLE_GRAPH: return "FORA_WHOLE_GRAPH";
    case AlgoType::FORA_TOPK: return "FORA_TOPK";
    case AlgoType::FWDPUSH: return "FWDPUSH";
    case AlgoType::MC: return "MC";
    default: return "BASE_WHOLE_GRAPH";
  }
}

Gen_Util::Gen_Util(std::shared_ptr<Graph> adjM, double alpha, std::string dir_db, uint64_t seed)
    : report_file(dir_db + "_AlgoPerfResults.txt"), adjM(std::move(adjM)), alpha(alpha), dir_db(dir_db), seed(seed) {}

std::vector<long> Gen_Util::getQueryNodes(int query_num) {  // Gen_Util.java:99-107 with a seeded generator
  std::vector<long> q;
  uint64_t s = seed ^ (0x51ED270B1A5ull + draws++);
  for (int i = 0; i < query_num; ++i)
    q.push_back((long)(((unsigned __int128)splitmix(s) * (uint64_t)adjM->nodeCount()) >> 64));
  return q;
}

double Gen_Util::maxErr(const PprMap& algo, const PprMap& gnd) {  // :306-321
  double m = 0.0;
  for (auto& e : gnd) {
    auto it = algo.find(e.first);
    m = std::max(m, std::fabs((it == algo.end() ? 0.0 : it->second) - e.second));
  }
  return m;
}
double Gen_Util::precision(const std::vector<long>& algo, const std::vector<long>& gnd) {  // :271-279
  double hit = 0.0;
  for (long a : algo)
    if (std::find(gnd.begin(), gnd.end(), a) != gnd.end()) hit++;
  return hit / (double)gnd.size();
}
double Gen_Util::ndcg(const std::vector<long>& algo, const std::vector<long>& gnd, const PprMap& gt) {  // :280-300
  double zk = 0.0, dcg = 0.0;
  for (size_t i = 1; i <= gt.size() && i <= gnd.size(); ++i)
    zk += (std::pow(2.0, gt.at(gnd[i - 1])) - 1.0) / std::log(i + 1.0) / std::log(2.0);
  for (size_t i = 1; i <= algo.size(); ++i) {
    auto it = gt.find(algo[i - 1]);
    dcg += (std::pow(2.0, it == gt.end() ? 0.0 : it->second) - 1.0) / std::log(i + 1.0) / std::log(2.0);
  }
  return dcg / zk;
}

void Gen_Util::algo_perf_test(AlgoType algoType, int query_num, int k, double param, double threshold,
                              bool to_be_preprocessed, TestType testType) {
  std::ofstream fw(report_file, std::ios::app);
  std::vector<long> queryNodes = getQueryNodes(query_num);
  Algo_Conf conf(alpha, seed);
  auto pm = conf.set_conf_power_method(adjM, dir_db);
  auto paramStr = [&](double p) { return p == (long)p && algoType == AlgoType::BASE_WHOLE_GRAPH ? std::to_string((long)p) : jdouble(p); };
  if (testType == TestType::TOPK) {  // :124-178
    std::unique_ptr<Topk_Util_Interface> topk;
    Base_Whole_Graph* base = nullptr;
    switch (algoType) {
      case AlgoType::FORA_TOPK: topk = conf.set_conf_fora_topk(k, adjM, dir_db); break;
      case AlgoType::FWDPUSH: topk = conf.set_conf_fwdpush(adjM, dir_db); break;
      case AlgoType::MC: topk = conf.set_conf_mc(adjM, dir_db); break;
      case AlgoType::BASE_WHOLE_GRAPH: {
        auto b = conf.set_conf_base_whole_graph(adjM, dir_db);
        base = b.get();
        topk = std::move(b);
      } break;
      default: throw PprError(PPRHIP_ERR_INVALID, "algo_perf_test: not a top-k algorithm");
    }
    double duration = 0.0, sum_precision = 0.0, sum_ndcg = 0.0;
    if (base) {
      const double t0 = now_ms();
      base->preprocessing(threshold, k);
      const double prep = now_ms() - t0;
      std::cout << "\nPreprocessing time for " << algoName(algoType) << ": " << (long)prep << "(ms)" << std::endl;
      fw << jdouble(threshold) << "," << k << "," << (long)prep << "," << base->getPrepSize() << ",";
    } else {
      fw << paramStr(param) << "," << k << ",";
    }
    std::cout << "\nTesting Top-k SSPPR performance of " << algoName(algoType) << " with " << query_num << " queries"
              << std::endl;
    for (int i = 0; i < query_num; ++i) {
      const double t0 = now_ms();
      topk->computeTopKPPR(queryNodes[i], k, param);
      duration += now_ms() - t0;
      std::vector<long> algo_ids = topk->getTopKNodeIds(k);
      pm->computeTopKPPR(queryNodes[i], k, 0.0);
      const std::vector<long>& gnd = pm->getTopKNodeIds(k);
      if (!gnd.empty()) {
        sum_precision += precision(algo_ids, gnd);
        sum_ndcg += ndcg(algo_ids, gnd, pm->getTopK(k));
      }
    }
    std::cout << "\nPerformance test of " << algoName(algoType) << " completed" << std::endl;
    const double avg_ms = duration / query_num;
    std::cout << "\n" << algoName(algoType) << " performance:\nAverage running time: " << avg_ms << "(ms)"
              << "\nAverage precision: " << jdouble(sum_precision / query_num)
              << "\nAverage NDCG: " << jdouble(sum_ndcg / query_num) << "\n" << std::endl;
    fw << (long)avg_ms << "," << jdouble(sum_precision / query_num) << "," << jdouble(sum_ndcg / query_num) << "\n";
    if (base) base->deletePrepDir();
    return;
  }
  // WHOLE_GRAPH (:180-252)
  std::unique_ptr<Whole_Graph_Util_Interface> algo;
  Preprocessing_Interface* prep = nullptr;
  switch (algoType) {
    case AlgoType::FORA_WHOLE_GRAPH: {
      auto a = conf.set_conf_fora_whole_graph(adjM, dir_db);
      prep = a.get();
      algo = std::move(a);
    } break;
    case AlgoType::FWDPUSH: {
      auto a = conf.set_conf_fwdpush(adjM, dir_db);
      prep = a.get();
      algo = std::move(a);
    } break;
    case AlgoType::MC: {
      auto a = conf.set_conf_mc(adjM, dir_db);
      prep = a.get();
      algo = std::move(a);
    } break;
    case AlgoType::BASE_WHOLE_GRAPH: {
      auto a = conf.set_conf_base_whole_graph(adjM, dir_db);
      prep = a.get();
      algo = std::move(a);
    } break;
    default: throw PprError(PPRHIP_ERR_INVALID, "algo_perf_test: not a whole-graph algorithm");
  }
  const bool is_base = algoType == AlgoType::BASE_WHOLE_GRAPH;
  fw << paramStr(param) << ",";
  const bool preprocessed = (to_be_preprocessed || is_base) && prep;
  if (preprocessed) {
    const double t0 = now_ms();
    prep->preprocessing(threshold, param);
    const double dur = now_ms() - t0;
    std::cout << "\nPreprocessing time for " << algoName(algoType) << ": " << (long)dur << "(ms)" << std::endl;
    fw << jdouble(threshold) << "," << (long)dur << "," << prep->getPrepSize() << ",";
  }
  std::cout << "\nTesting performance of " << algoName(algoType) << " with " << query_num << " queries" << std::endl;
  double duration = 0.0, sum_max_err = 0.0;
  // FORA's query loop (:208-232) goes to the GPU as one batch: the queries are known up front (:99-107)
  Fora_Whole_Graph* fora_batch =
      (algoType == AlgoType::FORA_WHOLE_GRAPH && !to_be_preprocessed) ? dynamic_cast<Fora_Whole_Graph*>(algo.get()) : nullptr;
  if (fora_batch) {
    const double t0 = now_ms();
    fora_batch->computeWholeGraphPPRBatch(queryNodes, param);
    duration = now_ms() - t0;
  }
  for (int i = 0; i < query_num; ++i) {
    const double t0 = now_ms();
    if (fora_batch)
      fora_batch->selectBatchResult((size_t)i);
    else if (to_be_preprocessed && prep)
      prep->readPreprocessedPPR(queryNodes[i]);
    else
      algo->computeWholeGraphPPR(queryNodes[i], param);
    if (!fora_batch) duration += now_ms() - t0;
    PprMap est = algo->getWholeGraphPPR();
    pm->computeWholeGraphPPR(queryNodes[i], 0.0);
    sum_max_err += maxErr(est, pm->getWholeGraphPPR());
  }
  std::cout << "\nPerformance test of " << algoName(algoType) << " completed" << std::endl;
  std::cout << "\n" << algoName(algoType) << " performance:\nAverage computing time: " << duration / query_num << "(ms)"
            << "\nAverage max error: " << jdouble(sum_max_err / query_num) << std::endl;
  if (!to_be_preprocessed) fw << (long)(duration / query_num) << ",";
  fw << jdouble(sum_max_err / query_num) << "\n";
  if (preprocessed) prep->deletePrepDir();
}

void Gen_Util::algo_perf_batch_test(int query_num, int k) {
  // Testset5 "parameters for GOT" (Gen_Util.java:451-478); the Neo4j built-in method is outside the hot path
  const std::vector<double> thr_base = {0.001, 5.0E-4, 5.0E-5, 1.0E-6, 5.0E-7};
  const std::vector<double> eps_fora = {10.0, 5.0, 0.5, 0.1, 0.05};
  const std::vector<double> eps_mc = {1.0, 0.5, 0.3, 0.1, 0.05};
  const std::vector<double> rmax_arr = {1.0E-4, 1.0E-5, 1.0E-6, 1.0E-7, 1.0E-8};
  {
    std::ofstream fw(report_file, std::ios::app);
    char ts[32];
    time_t now = time(nullptr);
    strftime(ts, sizeof ts, "%Y-%m-%d %H:%M:%S", localtime(&now));
    fw << ts << "\n\nTest 1. Whole-Graph test\n";
  }
  auto header = [&](const std::string& s) { std::ofstream(report_file, std::ios::app) << "\n" << s << "\n"; };
  const AlgoType whole[] = {AlgoType::FORA_WHOLE_GRAPH, AlgoType::FWDPUSH, AlgoType::MC, AlgoType::BASE_WHOLE_GRAPH};
  int idx = 1;
  for (AlgoType a : whole) {
    header("1." + std::to_string(idx++) + " " + algoName(a));
    if (a == AlgoType::BASE_WHOLE_GRAPH)
      for (double t : thr_base) algo_perf_test(a, query_num, -1, -1, t, false, TestType::WHOLE_GRAPH);
    else
      for (double p : (a == AlgoType::MC ? eps_mc : a == AlgoType::FWDPUSH ? rmax_arr : eps_fora))
        algo_perf_test(a, query_num, -1, p, -1.0, false, TestType::WHOLE_GRAPH);
  }
  header("Test 2. Top-k test");
  const AlgoType topk[] = {AlgoType::FORA_TOPK, AlgoType::FWDPUSH, AlgoType::MC, AlgoType::BASE_WHOLE_GRAPH};
  idx = 1;
  for (AlgoType a : topk) {
    header("2." + std::to_string(idx++) + " " + algoName(a));
    if (a == AlgoType::BASE_WHOLE_GRAPH)
      for (double t : thr_base) algo_perf_test(a, query_num, k, -1, t, false, TestType::TOPK);
    else
      for (double p : (a == AlgoType::MC ? eps_mc : a == AlgoType::FWDPUSH ? rmax_arr : eps_fora))
        algo_perf_test(a, query_num, k, p, -1.0, false, TestType::TOPK);
  }
  // Test 3. Preprocessing test (Gen_Util.java:602-645): every source's vector is computed up front and written to
  // <ALGO>_ppr_results/<db>/<param>/<id>.txt; a query then reads its file (readPreprocessedPPR).  Same parameter
  // arrays as Test 1 for GOT (:470-477); threshold_arr_other_prep = {-1.0}.
  header("Test 3. Preprocessing test");
  const AlgoType prep[] = {AlgoType::FORA_WHOLE_GRAPH, AlgoType::FWDPUSH, AlgoType::MC, AlgoType::BASE_WHOLE_GRAPH};
  idx = 1;
  for (AlgoType a : prep) {
    header("3." + std::to_string(idx++) + " " + algoName(a));
    if (a == AlgoType::BASE_WHOLE_GRAPH)
      for (double t : thr_base) algo_perf_test(a, query_num, -1, -1, t, true, TestType::WHOLE_GRAPH);
    else
      for (double p : (a == AlgoType::MC ? eps_mc : a == AlgoType::FWDPUSH ? rmax_arr : eps_fora))
        algo_perf_test(a, query_num, -1, p, -1.0, true, TestType::WHOLE_GRAPH);
  }
}

}  // namespace fora_neo4j
