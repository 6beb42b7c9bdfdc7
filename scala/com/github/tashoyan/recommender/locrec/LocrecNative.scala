package com.github.tashoyan.recommender.locrec

/**
  * JNI binding of liblocrec.so (include/locrec.h) through jni/locrec_jni.c.
  * Every method throws what the reference throws: IllegalArgumentException for a failed require()
  * or an unknown id, RuntimeException / OutOfMemoryError for device and allocation failures.
  * Handles are opaque Longs.  The operator classes do not own theirs: they take them from the library's
  * process-wide cache (LocrecBackend.cached) and release the reference in close() / through a Cleaner.
  *
  * Output convention (as in the header): the caller passes arrays; the return value is the number of
  * rows the result HAS - when it exceeds the arrays' length, call again with larger arrays.
  */
object LocrecNative {

  System.loadLibrary("locrec_jni") // liblocrec_jni.so links liblocrec.so ($ORIGIN rpath); -Djava.library.path=<dir>

  @native def version(): String

  @native def deviceCount(): Int

  @native def setDevice(ordinal: Int): Unit

  /** Default device list of the multi-device natives below (an empty array forgets it). */
  @native def setDevices(deviceIds: Array[Int]): Unit

  /** Bytes of device memory the library holds right now. */
  @native def deviceBytesInUse(): Long

  // ---- process-wide handle cache (include/locrec.h "Handle cache"); kind 0 = KNN index, 1 = SG graph
  /** The cached handle for `key` with a reference taken, or 0 on a miss. */
  @native def cacheAcquire(kind: Int, key: String): Long

  /** Hands a freshly created handle to the cache (which owns it from here on); returns the handle to use. */
  @native def cachePublish(kind: Int, key: String, handle: Long, deviceBytes: Long): Long

  /** Drops a reference (the handle stays cached); a handle that was never published is destroyed. */
  @native def cacheRelease(kind: Int, handle: Long): Unit

  /** -1 keeps a limit. */
  @native def cacheSetLimits(maxDeviceBytes: Long, maxEntries: Long): Unit

  /** out(0..4) = entries, entry bytes, hits, misses, evictions. */
  @native def cacheStats(out: Array[Long]): Unit

  // ---- KNN: knn/KnnRecommender.scala
  @native def knnCreate(
      personIds: Array[Long],
      pRowPtr: Array[Long], pIdx: Array[Int], pVal: Array[Double], pDim: Int,
      cRowPtr: Array[Long], cIdx: Array[Int], cVal: Array[Double], cDim: Int,
      rRowPtr: Array[Long], rPlace: Array[Long], rRating: Array[Long]
  ): Long

  @native def knnDestroy(handle: Long): Unit

  /** One replica of the index per device of `deviceIds` (null: the list of setDevices): the queries of
    * knnReplicasRecommendBatch are sharded over the devices (include/locrec.h "Several devices in one process"). */
  @native def knnReplicasCreate(
      deviceIds: Array[Int], personIds: Array[Long],
      pRowPtr: Array[Long], pIdx: Array[Int], pVal: Array[Double], pDim: Int,
      cRowPtr: Array[Long], cIdx: Array[Int], cVal: Array[Double], cDim: Int,
      rRowPtr: Array[Long], rPlace: Array[Long], rRating: Array[Long]
  ): Long

  @native def knnReplicasDestroy(handle: Long): Unit

  @native def knnReplicasRecommendBatch(
      handle: Long, personIds: Array[Long], placeWeight: Double, categoryWeight: Double, kNearest: Long,
      outOffsets: Array[Long], outPlaceIds: Array[Long], outEstimatedRatings: Array[Double]
  ): Long

  @native def knnRecommend(
      handle: Long, personId: Long, placeWeight: Double, categoryWeight: Double, kNearest: Long,
      outPlaceIds: Array[Long], outEstimatedRatings: Array[Double]
  ): Long

  @native def knnQuery(
      handle: Long, personId: Long, placeWeight: Double, categoryWeight: Double, kNearest: Long,
      outPersonIds: Array[Long], outSimilarities: Array[Double]
  ): Long

  @native def knnRecommendBatch(
      handle: Long, personIds: Array[Long], placeWeight: Double, categoryWeight: Double, kNearest: Long,
      outOffsets: Array[Long], outPlaceIds: Array[Long], outEstimatedRatings: Array[Double]
  ): Long

  // ---- SG: stochastic/StochasticRecommender.scala
  @native def sgCreate(sourceIds: Array[Long], targetIds: Array[Long], balancedWeights: Array[Double]): Long

  @native def sgDestroy(handle: Long): Unit

  @native def sgVertexCount(handle: Long): Long

  /** outIterationsConverged(0) = the 0-based counter the reference prints, (1) = 1 if converged. */
  @native def sgRecommend(
      handle: Long, vertexId: Long, alpha: Double, epsilon: Double, maxIterations: Long,
      outIds: Array[Long], outProbabilities: Array[Double], outIterationsConverged: Array[Long]
  ): Long

  /** One graph with its rows sharded over several devices; byTarget = rows of P^T, all-gather, bit-identical to one device. */
  @native def sgShardedCreate(deviceIds: Array[Int], sourceIds: Array[Long], targetIds: Array[Long], balancedWeights: Array[Double], byTarget: Boolean): Long

  @native def sgShardedDestroy(handle: Long): Unit

  @native def sgShardedVertexCount(handle: Long): Long

  @native def sgShardedRecommend(
      handle: Long, vertexId: Long, alpha: Double, epsilon: Double, maxIterations: Long,
      outIds: Array[Long], outProbabilities: Array[Double], outIterationsConverged: Array[Long]
  ): Long

  /** The result an *_async / group call left in one graph (rows, iteration counter, converged). */
  @native def sgFetch(handle: Long, outIds: Array[Long], outProbabilities: Array[Double], outIterationsConverged: Array[Long]): Long

  /** Many graphs (one per region and per region pair, StochasticRecommenderMain) iterated together. */
  @native def sgGroupCreate(graphHandles: Array[Long]): Long

  @native def sgGroupSweeps(group: Long, vertexIds: Array[Long], alpha: Double, sweeps: Long): Unit

  /** makeRecommendations' iteration (epsilon, maxIterations) for every graph; read each graph with sgFetch-style calls. */
  @native def sgGroupIterate(group: Long, vertexIds: Array[Long], alpha: Double, epsilon: Double, maxIterations: Long): Unit

  @native def sgGroupSynchronize(group: Long): Unit

  @native def sgGroupDestroy(group: Long): Unit

  // ---- Parquet sets -> device handles by native code (liblocrec_parquet.so, Arrow C++): no collect through the driver.
  // Paths are local directories (Spark output) or files; UnsupportedOperationException if the shim was built without it.
  @native def knnCreateFromParquet(placeRatingVectorsPath: String, categoryRatingVectorsPath: String, placeRatingsPath: String): Long

  @native def sgCreateFromParquet(stochasticGraphPath: String): Long

  // ---- the builders either side of the two recommenders (host arrays in and out)

  /** RatingsBuilder.calcRatings (knn/RatingsBuilder.scala:32-48); outputs of personIds.length entries; returns the row count. */
  @native def calcRatings(
      personIds: Array[Long], entityIds: Array[Long], topN: Long,
      outPersonIds: Array[Long], outEntityIds: Array[Long], outRatings: Array[Long]
  ): Long

  /** RatingVectorsBuilder.calcRatingVectors (:10-25,52-84); outCounts = (persons, non-zeros, vector size). */
  @native def calcRatingVectors(
      personIds: Array[Long], entityIds: Array[Long], ratings: Array[Long],
      outPersonIds: Array[Long], outRowPtr: Array[Long], outIdx: Array[Int], outVal: Array[Double], outCounts: Array[Long]
  ): Unit

  /** PlaceVisits.calcPlaceVisits (PlaceVisits.scala:11-46); returns the number of matches (may exceed the arrays' length). */
  @native def calcPlaceVisits(
      vPersonIds: Array[Long], vTimestamps: Array[Long], vLatitudes: Array[Double], vLongitudes: Array[Double], vRegionIds: Array[Long],
      pIds: Array[Long], pLatitudes: Array[Double], pLongitudes: Array[Double], pRegionIds: Array[Long], pCategoryIds: Array[Long],
      visitsFrom: Long, maxMeters: Double,
      outPersonIds: Array[Long], outTimestamps: Array[Long], outPlaceIds: Array[Long], outRegionIds: Array[Long], outCategoryIds: Array[Long]
  ): Long

  /** printRecommendations of both mains (knn/KnnRecommenderMain.scala:90-101); returns the row count. */
  @native def rankRecommendations(
      ids: Array[Long], scores: Array[Double], placeIds: Array[Long], placeRegionIds: Array[Long],
      targetRegionId: Long, maxRecommendations: Long, outIds: Array[Long], outScores: Array[Double]
  ): Long

}
