package joezie.fora_neo4j;

/**
 * JNI binding of libpprhip.so (include/pprhip.h): the MI355X engine behind the reference's algorithm classes.
 *
 * One instance = one graph replica on one GPU, used from one thread at a time (the reference's objects are
 * single-threaded too: Forward_Push.java:33-43 keeps per-query state in fields).  Results stay in HBM after a
 * compute call; reserve() / topk() fetch them.  Every native method throws RuntimeException with the engine's
 * message when the C ABI returns an error code; PPR.main already catches and prints those (PPR.java:196-199).
 *
 * How the reference's classes use it (INTEGRATION.md has the diffs):
 *   PPR.setupAdjMatrix           -> new PprHip(adjM, device)
 *   Fora_Whole_Graph.compute...  -> foraSingleSource + reserve()      (Fora_Whole_Graph.java:82-146)
 *   Gen_Util's query loop        -> foraBatch / foraBatchResident     (Gen_Util.java:208-232)
 *   ... called block after block  -> streamOpen / streamSubmit / streamWait / streamClose
 *   Fora_Topk.computeTopKPPR     -> foraTopk                          (Fora_Topk.java:102-199)
 *   Forward_Push                 -> forwardPush, fwdpushTopkReset/Round (Forward_Push.java:63-250)
 *   Monte_Carlo                  -> monteCarlo, randomWalks           (Monte_Carlo.java:60-158)
 *   Backward_Search              -> backwardPush                      (Backward_Search.java:38-100)
 *   Base_Whole_Graph.preprocessing -> allPairBackward (writes the same <id>.txt files, Base_Whole_Graph.java:58-164)
 *   Power_Method                 -> powerMethod                       (Power_Method.java:44-101)
 *
 * Not compiled in the engine's own image (no JDK there); build with `make jni JAVA_HOME=...`.
 */
public final class PprHip implements AutoCloseable {
    static {
        System.loadLibrary("pprhip_jni"); // links libpprhip.so
    }

    private long handle; // pprhip_graph_t*
    private long store;  // pprhip_results_t* of the last foraBatchResident call (0 = none)
    private long stream; // native box of the open query stream (0 = none)
    private final int n;

    /** Copies HeavyGraph's adjacency once (what PPR.setupAdjMatrix + set_configuration do, PPR.java:121-152). */
    public PprHip(org.neo4j.graphalgo.api.Graph adjM, int device) {
        n = (int) adjM.nodeCount();
        int[] outRp = new int[n + 1], inRp = new int[n + 1];
        for (int v = 0; v < n; v++) {
            outRp[v + 1] = outRp[v] + adjM.degree(v, org.neo4j.graphdb.Direction.OUTGOING);
            inRp[v + 1] = inRp[v] + adjM.degree(v, org.neo4j.graphdb.Direction.INCOMING);
        }
        int[] outCi = new int[outRp[n]], inCi = new int[inRp[n]];
        for (int v = 0; v < n; v++) {
            final int[] w = {outRp[v]}, x = {inRp[v]};
            adjM.forEachRelationship(v, org.neo4j.graphdb.Direction.OUTGOING, (a, b, r) -> { outCi[w[0]++] = b; return true; });
            adjM.forEachRelationship(v, org.neo4j.graphdb.Direction.INCOMING, (a, b, r) -> { inCi[x[0]++] = b; return true; });
        }
        handle = create(n, outRp, outCi, inRp, inCi, device);
    }

    public int nodeCount() { return n; }

    private static native long create(int n, int[] outRp, int[] outCi, int[] inRp, int[] inCi, int device);

    /** Fora_Whole_Graph.computeWholeGraphPPR; rounds = 0 lets the engine's cost model choose the halvings. */
    public native void foraSingleSource(int src, double eps, double alpha, long seed, int rounds);

    /** Gen_Util's query loop as one call, 16 queries in flight: fills idsOut/valsOut with q rows of k (id -1 pads). */
    public native void foraBatch(int[] srcs, double eps, double alpha, long seed, int k, int[] idsOut, double[] valsOut);

    /** The same with every query's whole-graph vector kept in HBM; batchResult(i) serves getWholeGraphPPR() of query i. */
    public native void foraBatchResident(int[] srcs, double eps, double alpha, long seed, int k, int[] idsOut, double[] valsOut);

    /** Vector of query i of the last foraBatchResident call (dense, mapped ids). */
    public native double[] batchResult(int i);

    /** Query stream (pprhip_fora_stream_*): blocks of sources submitted as they come; the engine starts a block's first
     *  queries in the slots the block before leaves, so short blocks (Gen_Util's 50) run at the rate of long ones.
     *  While a stream is open every other compute method of this object throws; streamClose() finishes what is queued. */
    public native void streamOpen(double eps, double alpha, int k);

    /** Queues a block; returns its ticket. */
    public native long streamSubmit(int[] srcs, long seed);

    /** Blocks until the block has finished and copies its q rows of k (id -1 pads) into idsOut / valsOut. */
    public native void streamWait(long ticket, int[] idsOut, double[] valsOut);

    public native void streamClose();

    /** Fora_Topk.computeTopKPPR + getTopKNodeIds: returns the number of entries >= the k-th value (may exceed k). */
    public native int foraTopk(int src, int k, double eps, double alpha, long seed, int[] idsOut, double[] valsOut);

    /** Forward_Push.computeWholeGraphPPR; returns rsum (exact sum of residues). */
    public native double forwardPush(int src, double alpha, double rmax);

    /** new Forward_Push(rsum = 1) + Q = {s} (Fora_Topk.java:117-118). */
    public native void fwdpushTopkReset(int src, double alpha);

    /** Forward_Push.forward_push_topk for one round; returns the updated rsum. */
    public native double fwdpushTopkRound(double minRmax, double rmax);

    /** Monte_Carlo.computeWholeGraphPPR. */
    public native void monteCarlo(int src, double eps, double alpha, long seed);

    /** Monte_Carlo.random_walk / random_walk_no_zero_hop for a batch of (start, walk index) pairs; returns terminals. */
    public native int[] randomWalks(int[] starts, long[] walkIdx, double alpha, long seed, int stream, boolean noZeroHop);

    /** Backward_Search.backward_search_whole_graph. */
    public native void backwardPush(int target, double alpha, double rmax);

    /** Power_Method.computeWholeGraphPPR (iters sweeps; the reference uses 100). */
    public native void powerMethod(int src, double alpha, int iters);

    /** Base_Whole_Graph.preprocessing(threshold, k): all targets, files "<id>.txt" under dir; returns the entry count. */
    public native long allPairBackward(double alpha, double threshold, int k, String dir);

    /** Result vector of the last single-query compute call (dense, mapped ids); entries > 0 are the reference's map. */
    public native double[] reserve();

    /** Residue vector of the last push. */
    public native double[] residue();

    /** Algo_Util.kth_ppr + retrieveTopK on the vector in HBM: returns the count selected, fills at most idsOut.length. */
    public native int topk(int k, int[] idsOut, double[] valsOut);

    /** Batched calls pay off with the batch cost-model profile (pprhip_tuning_batch); false restores the default. */
    public native void setBatchTuning(boolean on);

    /** Hands the workspaces of All-Pair (1) and / or of the batched calls (2) back to the device between the phases
     *  of a job (pprhip_graph_release); the next call of those entry points allocates them again. */
    public native void release(int what);

    @Override
    public native void close();

    /** The reference's HashMap view of a dense vector (keys = original ids, here equal to mapped ids). */
    public static java.util.HashMap<Long, Double> toMap(double[] dense) {
        java.util.HashMap<Long, Double> m = new java.util.HashMap<>();
        for (int v = 0; v < dense.length; v++) if (dense[v] > 0.0) m.put((long) v, dense[v]);
        return m;
    }
}
