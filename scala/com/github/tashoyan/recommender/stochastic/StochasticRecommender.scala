package com.github.tashoyan.recommender.stochastic

import com.github.tashoyan.recommender.locrec.{LocrecBackend, LocrecNative}
import org.apache.spark.sql.functions.col
import org.apache.spark.sql.types.{DoubleType, LongType, StructField, StructType}
import org.apache.spark.sql.{DataFrame, Row, SparkSession}

/**
  * Drop-in replacement of the reference class of the same name and package
  * (recommender/src/main/scala/com/github/tashoyan/recommender/stochastic/StochasticRecommender.scala:28-34,66-71):
  * same constructor (including the implicit session), same require()s, same method, same output
  * columns `(id: Long, probability: Double)`, the same two progress lines on Console.out
  * (:94,100).  The edge list is collected once PER PROCESS AND GRAPH FILE: StochasticRecommenderMain
  * (StochasticRecommenderMain.scala:53-62) constructs a new recommender for every request and never closes it,
  * so the device graph comes from the library's process-wide cache keyed by the frame's input files
  * (LocrecBackend.frameKey); epsilon and maxIterations are arguments of the native call, not part of the key.
  * Each request is one native call that runs step() (:92-106) on the device.
  * `LOCREC_BACKEND=spark` delegates to the reference's implementation (renamed SparkStochasticRecommender).
  */
class StochasticRecommender(
    stochasticEdges: DataFrame,
    epsilon: Double,
    maxIterations: Int
)(implicit spark: SparkSession) extends AutoCloseable {
  require(epsilon >= 0, "epsilon must be non-negative")
  require(maxIterations >= 0, "max iterations number must be non-negative")

  private val alpha: Double = 0.15 // StochasticRecommender.scala:38

  private val outputSchema = StructType(Seq(
    StructField("id", LongType, nullable = false),
    StructField("probability", DoubleType, nullable = false)
  ))

  private lazy val sparkDelegate = new SparkStochasticRecommender(stochasticEdges, epsilon, maxIterations)

  private lazy val (handle: Long, cleanable: java.lang.ref.Cleaner.Cleanable) =
    LocrecBackend.handleFor(this, LocrecBackend.KindSg, LocrecBackend.frameKey(stochasticEdges))(createGraph())

  private def createGraph(): Long = {
    LocrecBackend.localPathOf(stochasticEdges).flatMap(p => LocrecBackend.tryNativeLoad(LocrecNative.sgCreateFromParquet(p))) match {
      case Some(h) => return h
      case None =>
    }
    // ids may arrive as Int and are widened (StochasticGraphBuilderTest.scala:20-23,56)
    val edges = stochasticEdges
      .select(col("source_id").cast(LongType), col("target_id").cast(LongType), col("balanced_weight").cast(DoubleType))
      .collect()
    val source = new Array[Long](edges.length)
    val target = new Array[Long](edges.length)
    val weight = new Array[Double](edges.length)
    var i = 0
    while (i < edges.length) {
      source(i) = edges(i).getLong(0)
      target(i) = edges(i).getLong(1)
      weight(i) = edges(i).getDouble(2)
      i += 1
    }
    LocrecNative.sgCreate(source, target, weight)
  }

  def makeRecommendations(vertexId: Long): DataFrame = if (LocrecBackend.useSpark) sparkDelegate.makeRecommendations(vertexId) else LocrecBackend.lockOf(handle).synchronized {
    val capacity = math.max(LocrecNative.sgVertexCount(handle), 1L).toInt
    val ids = new Array[Long](capacity)
    val probabilities = new Array[Double](capacity)
    val iterationsConverged = new Array[Long](2)
    // throws IllegalArgumentException(s"No such vertex in the graph: $vertexId") exactly as :70
    val count = LocrecNative.sgRecommend(handle, vertexId, alpha, epsilon, maxIterations.toLong, ids, probabilities, iterationsConverged)
    if (iterationsConverged(1) != 0L)
      Console.out.println(s"Converged in ${iterationsConverged(0)} iterations")
    else
      Console.out.println(s"Number of iterations ${iterationsConverged(0)} reached the maximum $maxIterations")
    val rows = (0 until count.toInt).map(i => Row(ids(i), probabilities(i)))
    spark.createDataFrame(spark.sparkContext.parallelize(rows, 1), outputSchema)
  }

  /** Drops this object's reference (idempotent); the device graph stays cached for the next constructor. */
  override def close(): Unit = if (!LocrecBackend.useSpark) cleanable.clean()

}
