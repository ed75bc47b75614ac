// pprhip_jni.cpp — JNI side of joezie.fora_neo4j.PprHip: every native method is one call of the C ABI
// (include/pprhip.h).  Build where a JDK exists:
//   make -C personalized-pagerank-algorithms-on-neo4j_amd jni JAVA_HOME=/path/to/jdk
// This image has none, so `make jni-check` (part of `make all`) only type-checks this file against
// host/jni/stub/jni.h, a declaration-only stand-in for the calls used here, so that the file cannot rot unseen.
#include <jni.h>

#include <cstring>
#include <map>
#include <vector>

#include "../../../include/pprhip.h"

namespace {

// field ids of PprHip.handle / PprHip.store, looked up once (they stay valid while the class is loaded; two threads
// racing here store the same values)
struct FieldIds {
  jfieldID handle = nullptr, store = nullptr, stream = nullptr;
};
FieldIds& fields(JNIEnv* e, jobject self) {
  static FieldIds F;
  if (!F.handle || !F.store || !F.stream) {
    jclass c = e->GetObjectClass(self);
    F.handle = e->GetFieldID(c, "handle", "J");
    F.store = e->GetFieldID(c, "store", "J");
    F.stream = e->GetFieldID(c, "stream", "J");
  }
  return F;
}
jfieldID fid_store(JNIEnv* e, jobject self) { return fields(e, self).store; }
jfieldID fid_handle(JNIEnv* e, jobject self) { return fields(e, self).handle; }

bool fail(JNIEnv* e, int rc);

void throw_new(JNIEnv* e, const char* cls, const char* msg) {
  jclass c = e->FindClass(cls);
  if (c) e->ThrowNew(c, msg);
}

// the object's graph handle; a closed object (handle 0) raises IllegalStateException and yields nullptr
pprhip_graph_t* G(JNIEnv* e, jobject self) {
  const jfieldID f = fid_handle(e, self);
  pprhip_graph_t* g = f ? (pprhip_graph_t*)e->GetLongField(self, f) : nullptr;
  if (!g && !e->ExceptionCheck()) throw_new(e, "java/lang/IllegalStateException", "PprHip: the graph handle is closed");
  return g;
}
// An open query stream and the output buffers of its submissions: the library writes a submission's top-k block until
// its wait returns, so the block lives here, not in a Java array that the collector may move.
struct StreamBox {
  pprhip_stream_t* s = nullptr;
  int k = 0;
  struct Block {
    std::vector<int32_t> ids;
    std::vector<double> vals;
  };
  std::map<uint64_t, Block*> blocks;
};
StreamBox* SB(JNIEnv* e, jobject self) {
  const jfieldID f = fields(e, self).stream;
  StreamBox* b = f ? (StreamBox*)e->GetLongField(self, f) : nullptr;
  if (!b && !e->ExceptionCheck()) throw_new(e, "java/lang/IllegalStateException", "PprHip: no query stream is open");
  return b;
}
void close_stream(JNIEnv* e, jobject self, bool report) {
  const jfieldID f = fields(e, self).stream;
  StreamBox* b = f ? (StreamBox*)e->GetLongField(self, f) : nullptr;
  if (!b) return;
  e->SetLongField(self, f, 0);
  const int rc = pprhip_fora_stream_close(b->s);  // finishes everything submitted: nothing writes the blocks any more
  for (auto& kv : b->blocks) delete kv.second;
  delete b;
  if (report) fail(e, rc);
}

pprhip_results_t* R(JNIEnv* e, jobject self) {
  const jfieldID f = fid_store(e, self);
  return f ? (pprhip_results_t*)e->GetLongField(self, f) : nullptr;
}

// error code -> RuntimeException carrying the engine's message; returns true when an exception is pending
bool fail(JNIEnv* e, int rc) {
  if (rc == PPRHIP_OK) return false;
  throw_new(e, "java/lang/RuntimeException", pprhip_last_error());
  return true;
}

struct Dims {
  uint32_t n = 0;
  uint64_t m = 0;
};
Dims dims(pprhip_graph_t* g) {
  Dims d;
  pprhip_graph_info(g, &d.n, &d.m, nullptr);
  return d;
}

jdoubleArray dense(JNIEnv* e, pprhip_graph_t* g, int (*get)(pprhip_graph_t*, double*)) {
  if (!g) return nullptr;
  const Dims d = dims(g);
  std::vector<double> v(d.n);
  if (fail(e, get(g, v.data()))) return nullptr;
  jdoubleArray out = e->NewDoubleArray((jsize)d.n);
  e->SetDoubleArrayRegion(out, 0, (jsize)d.n, v.data());
  return out;
}

void batch(JNIEnv* e, jobject self, jintArray srcs, jdouble eps, jdouble alpha, jlong seed, jint k, jintArray idsOut,
           jdoubleArray valsOut, bool resident) {
  pprhip_graph_t* g = G(e, self);
  if (!g) return;
  if (!srcs || k < 0 || (k > 0 && (!idsOut || !valsOut))) {
    throw_new(e, "java/lang/IllegalArgumentException", "PprHip.foraBatch: null array or negative k");
    return;
  }
  const Dims d = dims(g);
  pprhip_fora_conf_t conf;
  if (fail(e, pprhip_conf_fora_whole_graph(d.n, d.m, alpha, &conf))) return;  // Algo_Conf.java:45-53
  const jsize q = e->GetArrayLength(srcs);
  if (k > 0 && ((jlong)e->GetArrayLength(idsOut) < (jlong)q * k || (jlong)e->GetArrayLength(valsOut) < (jlong)q * k)) {
    throw_new(e, "java/lang/IllegalArgumentException", "PprHip.foraBatch: output arrays shorter than q * k");
    return;
  }
  std::vector<jint> s(q);
  e->GetIntArrayRegion(srcs, 0, q, s.data());
  std::vector<int32_t> ids((size_t)q * k);
  std::vector<double> vals((size_t)q * k);
  pprhip_results_t* keep = nullptr;
  if (resident) {
    keep = R(e, self);
    int cap = 0;
    if (keep) pprhip_results_info(keep, &cap, nullptr, nullptr);
    if (!keep || cap < q) {  // (re)size the store to this batch
      if (keep) pprhip_results_destroy(keep);
      keep = nullptr;
      e->SetLongField(self, fid_store(e, self), 0);
      if (fail(e, pprhip_results_create(g, q, &keep))) return;
      e->SetLongField(self, fid_store(e, self), (jlong)keep);
    }
  }
  if (fail(e, pprhip_fora_batch_single_source_resident(g, (const int32_t*)s.data(), q, eps, &conf, (uint64_t)seed, 0, keep,
                                                       nullptr, k, ids.data(), vals.data(), nullptr, nullptr, nullptr)))
    return;
  e->SetIntArrayRegion(idsOut, 0, (jsize)ids.size(), (const jint*)ids.data());
  e->SetDoubleArrayRegion(valsOut, 0, (jsize)vals.size(), vals.data());
}

}  // namespace

extern "C" {

JNIEXPORT jlong JNICALL Java_joezie_fora_1neo4j_PprHip_create(JNIEnv* e, jclass, jint n, jintArray outRp, jintArray outCi,
                                                              jintArray inRp, jintArray inCi, jint device) {
  // the arrays are the caller's: everything the engine will index with is checked against their lengths first
  if (n < 1 || !outRp || !outCi || !inRp || !inCi) {
    throw_new(e, "java/lang/IllegalArgumentException", "PprHip.create: n < 1 or a null array");
    return 0;
  }
  if ((jlong)e->GetArrayLength(outRp) < (jlong)n + 1 || (jlong)e->GetArrayLength(inRp) < (jlong)n + 1) {
    throw_new(e, "java/lang/IllegalArgumentException", "PprHip.create: a row-pointer array is shorter than n + 1");
    return 0;
  }
  jint m = 0, m_in = 0;
  e->GetIntArrayRegion(outRp, n, 1, &m);
  e->GetIntArrayRegion(inRp, n, 1, &m_in);
  if (m < 0 || m_in != m || e->GetArrayLength(outCi) < m || e->GetArrayLength(inCi) < m) {
    throw_new(e, "java/lang/IllegalArgumentException",
              "PprHip.create: edge count (row_ptr[n]) negative, different on the two sides, or beyond a column array");
    return 0;
  }
  jint *orp = e->GetIntArrayElements(outRp, nullptr), *oci = e->GetIntArrayElements(outCi, nullptr);
  jint *irp = e->GetIntArrayElements(inRp, nullptr), *ici = e->GetIntArrayElements(inCi, nullptr);
  pprhip_graph_t* g = nullptr;
  int rc = PPRHIP_ERR_OOM;
  if (orp && oci && irp && ici)
    rc = pprhip_graph_create((uint32_t)n, (uint64_t)m, (const uint32_t*)orp, oci, (const uint32_t*)irp, ici, device, &g);
  if (orp) e->ReleaseIntArrayElements(outRp, orp, JNI_ABORT);
  if (oci) e->ReleaseIntArrayElements(outCi, oci, JNI_ABORT);
  if (irp) e->ReleaseIntArrayElements(inRp, irp, JNI_ABORT);
  if (ici) e->ReleaseIntArrayElements(inCi, ici, JNI_ABORT);
  if (!(orp && oci && irp && ici)) {
    if (!e->ExceptionCheck()) throw_new(e, "java/lang/OutOfMemoryError", "PprHip.create: array elements unavailable");
    return 0;
  }
  fail(e, rc);
  return (jlong)g;
}

JNIEXPORT void JNICALL Java_joezie_fora_1neo4j_PprHip_foraSingleSource(JNIEnv* e, jobject self, jint src, jdouble eps,
                                                                       jdouble alpha, jlong seed, jint rounds) {
  pprhip_graph_t* g = G(e, self);
  if (!g) return;
  const Dims d = dims(g);
  pprhip_fora_conf_t conf;
  if (fail(e, pprhip_conf_fora_whole_graph(d.n, d.m, alpha, &conf))) return;  // Algo_Conf.java:45-53
  fail(e, pprhip_fora_single_source(g, src, eps, &conf, (uint64_t)seed, rounds, nullptr, nullptr));
}

JNIEXPORT void JNICALL Java_joezie_fora_1neo4j_PprHip_foraBatch(JNIEnv* e, jobject self, jintArray srcs, jdouble eps,
                                                                jdouble alpha, jlong seed, jint k, jintArray idsOut,
                                                                jdoubleArray valsOut) {
  batch(e, self, srcs, eps, alpha, seed, k, idsOut, valsOut, false);
}

JNIEXPORT void JNICALL Java_joezie_fora_1neo4j_PprHip_foraBatchResident(JNIEnv* e, jobject self, jintArray srcs,
                                                                        jdouble eps, jdouble alpha, jlong seed, jint k,
                                                                        jintArray idsOut, jdoubleArray valsOut) {
  batch(e, self, srcs, eps, alpha, seed, k, idsOut, valsOut, true);
}

JNIEXPORT jdoubleArray JNICALL Java_joezie_fora_1neo4j_PprHip_batchResult(JNIEnv* e, jobject self, jint i) {
  pprhip_results_t* r = R(e, self);
  uint32_t n = 0;
  if (!r || fail(e, pprhip_results_info(r, nullptr, nullptr, &n))) {
    if (!r) throw_new(e, "java/lang/IllegalStateException", "no foraBatchResident call yet");
    return nullptr;
  }
  std::vector<double> v(n);
  if (fail(e, pprhip_results_fetch(r, i, v.data()))) return nullptr;
  jdoubleArray out = e->NewDoubleArray((jsize)n);
  e->SetDoubleArrayRegion(out, 0, (jsize)n, v.data());
  return out;
}

JNIEXPORT jint JNICALL Java_joezie_fora_1neo4j_PprHip_foraTopk(JNIEnv* e, jobject self, jint src, jint k, jdouble eps,
                                                               jdouble alpha, jlong seed, jintArray idsOut,
                                                               jdoubleArray valsOut) {
  pprhip_graph_t* g = G(e, self);
  if (!g) return 0;
  if (!idsOut || !valsOut || e->GetArrayLength(valsOut) < e->GetArrayLength(idsOut)) {
    throw_new(e, "java/lang/IllegalArgumentException", "PprHip.foraTopk: null output array, or values shorter than ids");
    return 0;
  }
  const Dims d = dims(g);
  pprhip_fora_conf_t conf;
  if (fail(e, pprhip_conf_fora_topk(d.n, d.m, k, alpha, &conf))) return 0;  // Algo_Conf.java:71-81
  const jsize cap = e->GetArrayLength(idsOut);
  std::vector<int32_t> ids(cap > 0 ? cap : 1);
  std::vector<double> vals(cap > 0 ? cap : 1);
  int n_sel = 0;
  if (fail(e, pprhip_fora_topk(g, src, eps, &conf, (uint64_t)seed, ids.data(), vals.data(), cap, &n_sel, nullptr, nullptr)))
    return 0;
  const jsize w = n_sel < cap ? n_sel : cap;
  e->SetIntArrayRegion(idsOut, 0, w, (const jint*)ids.data());
  e->SetDoubleArrayRegion(valsOut, 0, w, vals.data());
  return n_sel;
}

JNIEXPORT jdouble JNICALL Java_joezie_fora_1neo4j_PprHip_forwardPush(JNIEnv* e, jobject self, jint src, jdouble alpha,
                                                                     jdouble rmax) {
  double rsum = 0.0;
  pprhip_graph_t* g = G(e, self);
  if (g) fail(e, pprhip_forward_push(g, src, alpha, rmax, nullptr, nullptr, &rsum, nullptr));
  return rsum;
}

JNIEXPORT void JNICALL Java_joezie_fora_1neo4j_PprHip_fwdpushTopkReset(JNIEnv* e, jobject self, jint src, jdouble alpha) {
  pprhip_graph_t* g = G(e, self);
  if (g) fail(e, pprhip_fwdpush_topk_reset(g, src, alpha));
}

JNIEXPORT jdouble JNICALL Java_joezie_fora_1neo4j_PprHip_fwdpushTopkRound(JNIEnv* e, jobject self, jdouble minRmax,
                                                                          jdouble rmax) {
  double rsum = 0.0;
  pprhip_graph_t* g = G(e, self);
  if (g) fail(e, pprhip_fwdpush_topk_round(g, minRmax, rmax, &rsum, nullptr));
  return rsum;
}

JNIEXPORT void JNICALL Java_joezie_fora_1neo4j_PprHip_monteCarlo(JNIEnv* e, jobject self, jint src, jdouble eps,
                                                                 jdouble alpha, jlong seed) {
  pprhip_graph_t* g = G(e, self);
  if (!g) return;
  const Dims d = dims(g);
  pprhip_fora_conf_t conf;
  if (fail(e, pprhip_conf_fora_whole_graph(d.n, d.m, alpha, &conf))) return;  // Algo_Conf.java:29-35: same delta, pfail
  fail(e, pprhip_monte_carlo(g, src, eps, &conf, (uint64_t)seed, nullptr, nullptr));
}

JNIEXPORT jintArray JNICALL Java_joezie_fora_1neo4j_PprHip_randomWalks(JNIEnv* e, jobject self, jintArray starts,
                                                                       jlongArray walkIdx, jdouble alpha, jlong seed,
                                                                       jint stream, jboolean noZeroHop) {
  pprhip_graph_t* g = G(e, self);
  if (!g) return nullptr;
  if (!starts || !walkIdx || e->GetArrayLength(walkIdx) < e->GetArrayLength(starts)) {
    throw_new(e, "java/lang/IllegalArgumentException", "PprHip.randomWalks: null array, or walkIdx shorter than starts");
    return nullptr;
  }
  const jsize c = e->GetArrayLength(starts);
  std::vector<jint> s(c);
  std::vector<jlong> ix(c);
  e->GetIntArrayRegion(starts, 0, c, s.data());
  e->GetLongArrayRegion(walkIdx, 0, c, ix.data());
  std::vector<int32_t> term(c > 0 ? c : 1);
  if (fail(e, pprhip_random_walk_batch(g, (const int32_t*)s.data(), (const uint64_t*)ix.data(), (uint64_t)c, alpha,
                                       (uint64_t)seed, (uint32_t)stream, noZeroHop ? 1 : 0, term.data(), nullptr)))
    return nullptr;
  jintArray out = e->NewIntArray(c);
  e->SetIntArrayRegion(out, 0, c, (const jint*)term.data());
  return out;
}

JNIEXPORT void JNICALL Java_joezie_fora_1neo4j_PprHip_backwardPush(JNIEnv* e, jobject self, jint target, jdouble alpha,
                                                                   jdouble rmax) {
  pprhip_graph_t* g = G(e, self);
  if (g) fail(e, pprhip_backward_push(g, target, alpha, rmax, nullptr, nullptr, nullptr));
}

JNIEXPORT void JNICALL Java_joezie_fora_1neo4j_PprHip_powerMethod(JNIEnv* e, jobject self, jint src, jdouble alpha,
                                                                  jint iters) {
  pprhip_graph_t* g = G(e, self);
  if (g) fail(e, pprhip_power_method(g, src, alpha, iters, nullptr, nullptr));
}

JNIEXPORT jlong JNICALL Java_joezie_fora_1neo4j_PprHip_allPairBackward(JNIEnv* e, jobject self, jdouble alpha,
                                                                       jdouble threshold, jint k, jstring dir) {
  pprhip_graph_t* g = G(e, self);
  if (!g) return 0;
  if (!dir) {
    throw_new(e, "java/lang/IllegalArgumentException", "PprHip.allPairBackward: null directory");
    return 0;
  }
  const Dims d = dims(g);
  pprhip_index_t* ix = nullptr;
  if (fail(e, pprhip_all_pair_backward(g, alpha, threshold, k, 0, d.n, &ix, nullptr))) return 0;
  const char* path = e->GetStringUTFChars(dir, nullptr);
  if (!path) {  // OutOfMemoryError is pending
    pprhip_index_destroy(ix);
    return 0;
  }
  const int rc = pprhip_index_write_dir(ix, path);  // "<t>\t<Double.toString>\n" files (Base_Whole_Graph.java:118-126)
  e->ReleaseStringUTFChars(dir, path);
  uint64_t entries = 0;
  pprhip_index_info(ix, nullptr, &entries);
  pprhip_index_destroy(ix);
  fail(e, rc);
  return (jlong)entries;
}

JNIEXPORT jdoubleArray JNICALL Java_joezie_fora_1neo4j_PprHip_reserve(JNIEnv* e, jobject self) {
  return dense(e, G(e, self), pprhip_get_reserve);
}

JNIEXPORT jdoubleArray JNICALL Java_joezie_fora_1neo4j_PprHip_residue(JNIEnv* e, jobject self) {
  return dense(e, G(e, self), pprhip_get_residue);
}

JNIEXPORT jint JNICALL Java_joezie_fora_1neo4j_PprHip_topk(JNIEnv* e, jobject self, jint k, jintArray idsOut,
                                                           jdoubleArray valsOut) {
  pprhip_graph_t* g = G(e, self);
  if (!g) return 0;
  if (!idsOut || !valsOut || e->GetArrayLength(valsOut) < e->GetArrayLength(idsOut)) {
    throw_new(e, "java/lang/IllegalArgumentException", "PprHip.topk: null output array, or values shorter than ids");
    return 0;
  }
  const jsize cap = e->GetArrayLength(idsOut);
  std::vector<int32_t> ids(cap > 0 ? cap : 1);
  std::vector<double> vals(cap > 0 ? cap : 1);
  int n_sel = 0;
  if (fail(e, pprhip_topk_select(g, k, ids.data(), vals.data(), cap, &n_sel, nullptr, nullptr))) return 0;
  const jsize w = n_sel < cap ? n_sel : cap;
  e->SetIntArrayRegion(idsOut, 0, w, (const jint*)ids.data());
  e->SetDoubleArrayRegion(valsOut, 0, w, vals.data());
  return n_sel;
}

JNIEXPORT void JNICALL Java_joezie_fora_1neo4j_PprHip_setBatchTuning(JNIEnv* e, jobject self, jboolean on) {
  pprhip_tuning_t t;
  if (on) pprhip_tuning_batch(&t);
  else pprhip_tuning_default(&t);
  pprhip_graph_t* g = G(e, self);
  if (g) fail(e, pprhip_graph_set_tuning(g, &t));
}

JNIEXPORT void JNICALL Java_joezie_fora_1neo4j_PprHip_release(JNIEnv* e, jobject self, jint what) {
  pprhip_graph_t* g = G(e, self);
  if (g) fail(e, pprhip_graph_release(g, (unsigned)what));
}

// ---- query stream (pprhip_fora_stream_*): Gen_Util's loop called again and again without a drain between the calls
JNIEXPORT void JNICALL Java_joezie_fora_1neo4j_PprHip_streamOpen(JNIEnv* e, jobject self, jdouble eps, jdouble alpha, jint k) {
  pprhip_graph_t* g = G(e, self);
  if (!g) return;
  const jfieldID f = fields(e, self).stream;
  if (!f || e->GetLongField(self, f)) {
    throw_new(e, "java/lang/IllegalStateException", "PprHip.streamOpen: a stream is already open");
    return;
  }
  const Dims d = dims(g);
  pprhip_fora_conf_t conf;
  if (fail(e, pprhip_conf_fora_whole_graph(d.n, d.m, alpha, &conf))) return;  // Algo_Conf.java:45-53
  StreamBox* b = new StreamBox();
  b->k = k;
  if (fail(e, pprhip_fora_stream_open(g, eps, &conf, k, &b->s))) {
    delete b;
    return;
  }
  e->SetLongField(self, f, (jlong)b);
}

JNIEXPORT jlong JNICALL Java_joezie_fora_1neo4j_PprHip_streamSubmit(JNIEnv* e, jobject self, jintArray srcs, jlong seed) {
  StreamBox* b = SB(e, self);
  if (!b) return 0;
  if (!srcs) {
    throw_new(e, "java/lang/IllegalArgumentException", "PprHip.streamSubmit: null sources");
    return 0;
  }
  const jsize q = e->GetArrayLength(srcs);
  std::vector<jint> s(q);
  e->GetIntArrayRegion(srcs, 0, q, s.data());
  StreamBox::Block* blk = new StreamBox::Block();
  blk->ids.resize((size_t)q * b->k);
  blk->vals.resize((size_t)q * b->k);
  uint64_t ticket = 0;
  if (fail(e, pprhip_fora_stream_submit(b->s, (const int32_t*)s.data(), q, (uint64_t)seed, nullptr, 0, blk->ids.data(),
                                        blk->vals.data(), nullptr, &ticket))) {
    delete blk;
    return 0;
  }
  b->blocks[ticket] = blk;
  return (jlong)ticket;
}

JNIEXPORT void JNICALL Java_joezie_fora_1neo4j_PprHip_streamWait(JNIEnv* e, jobject self, jlong ticket, jintArray idsOut,
                                                                 jdoubleArray valsOut) {
  StreamBox* b = SB(e, self);
  if (!b) return;
  auto it = b->blocks.find((uint64_t)ticket);
  if (it == b->blocks.end()) {
    throw_new(e, "java/lang/IllegalArgumentException", "PprHip.streamWait: no open submission with this ticket");
    return;
  }
  StreamBox::Block* blk = it->second;
  if (b->k > 0 && (!idsOut || !valsOut || (size_t)e->GetArrayLength(idsOut) < blk->ids.size() ||
                   (size_t)e->GetArrayLength(valsOut) < blk->vals.size())) {
    throw_new(e, "java/lang/IllegalArgumentException", "PprHip.streamWait: output arrays shorter than q * k");
    return;
  }
  const int rc = pprhip_fora_stream_wait(b->s, (uint64_t)ticket, nullptr);
  b->blocks.erase(it);
  if (!fail(e, rc) && b->k > 0) {
    e->SetIntArrayRegion(idsOut, 0, (jsize)blk->ids.size(), (const jint*)blk->ids.data());
    e->SetDoubleArrayRegion(valsOut, 0, (jsize)blk->vals.size(), blk->vals.data());
  }
  delete blk;
}

JNIEXPORT void JNICALL Java_joezie_fora_1neo4j_PprHip_streamClose(JNIEnv* e, jobject self) { close_stream(e, self, true); }

JNIEXPORT void JNICALL Java_joezie_fora_1neo4j_PprHip_close(JNIEnv* e, jobject self) {
  // closing twice is allowed (AutoCloseable): the second call finds every field 0
  close_stream(e, self, false);
  if (pprhip_results_t* r = R(e, self)) pprhip_results_destroy(r);
  if (fid_store(e, self)) e->SetLongField(self, fid_store(e, self), 0);
  const jfieldID fh = fid_handle(e, self);
  if (!fh) return;
  pprhip_graph_destroy((pprhip_graph_t*)e->GetLongField(self, fh));
  e->SetLongField(self, fh, 0);
}

}  // extern "C"
