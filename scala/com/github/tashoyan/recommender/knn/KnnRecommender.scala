package com.github.tashoyan.recommender.knn

import com.github.tashoyan.recommender.locrec.{LocrecBackend, LocrecNative}
import org.apache.spark.ml.linalg.SparseVector
import org.apache.spark.sql.functions.col
import org.apache.spark.sql.types.{DoubleType, LongType, StructField, StructType}
import org.apache.spark.sql.{DataFrame, Row, SparkSession}

import scala.collection.mutable

/**
  * Drop-in replacement of the reference class of the same name and package
  * (recommender/src/main/scala/com/github/tashoyan/recommender/knn/KnnRecommender.scala:9-25):
  * same constructor, same require()s and messages, same method, same output columns
  * `(place_id: Long, estimated_rating: Double)`.  The body collects the three DataFrames to CSR
  * arrays ONCE PER PROCESS AND DATA SET and hands them to liblocrec.so; each request is then one native
  * call.  Callers (KnnRecommenderMain.makeRecommendations, KnnRecommenderMain.scala:53-67) need no change:
  * that main constructs a new KnnRecommender from freshly read DataFrames for every request and never closes
  * it, so the device index is taken from the library's process-wide cache, keyed by the three frames' input
  * files (LocrecBackend.framesKey) - a second request for the same region pair collects nothing and builds
  * nothing.  Weights and K are arguments of every native call, not part of the key.
  *
  * `LOCREC_BACKEND=spark` delegates every call to the reference's own implementation (renamed
  * SparkKnnRecommender, INTEGRATION.md section 3) for A/B runs.
  *
  * Tie order of orderBy(desc).limit(K) is undefined in Spark; the device defines it as
  * (similarity desc, person_id asc) - DESIGN.md section 2.
  */
class KnnRecommender(
    placeRatingVectors: DataFrame,
    categoryRatingVectors: DataFrame,
    placeRatings: DataFrame,
    placeWeight: Double,
    categoryWeight: Double,
    kNearest: Int
) extends AutoCloseable {
  require(placeWeight > 0 && placeWeight < 1.0, s"Place weight must be in the interval (0; 1): $placeWeight")
  require(categoryWeight > 0 && categoryWeight < 1.0, s"Category weight must be in the interval (0; 1): $categoryWeight")
  require(placeWeight + categoryWeight == 1.0, s"Sum of weights must be 1.0: place: $placeWeight, category: $categoryWeight")
  require(kNearest > 0, "K nearest must be positive")

  private val spark: SparkSession = placeRatingVectors.sparkSession

  private val outputSchema = StructType(Seq(
    StructField("place_id", LongType, nullable = false),
    StructField("estimated_rating", DoubleType, nullable = false)
  ))

  /** (person_id, rating_vector) rows -> person -> vector, plus the common vector size. */
  private def collectVectors(df: DataFrame): (Map[Long, SparseVector], Int) = {
    val rows = df.select(col("person_id").cast(LongType), col("rating_vector")).collect()
    val vectors = rows.map(r => r.getLong(0) -> r.getAs[SparseVector](1)).toMap
    val sizes = vectors.values.map(_.size).toSet
    require(sizes.size <= 1, s"Rating vectors of different sizes: ${sizes.toSeq.sorted.mkString(", ")}")
    (vectors, sizes.headOption.getOrElse(1))
  }

  /** CSR of one family over the given person order; a person absent from the frame gets an empty row. */
  private def toCsr(persons: Array[Long], vectors: Map[Long, SparseVector]): (Array[Long], Array[Int], Array[Double]) = {
    val rowPtr = new Array[Long](persons.length + 1)
    val idx = mutable.ArrayBuilder.make[Int]
    val value = mutable.ArrayBuilder.make[Double]
    var i = 0
    while (i < persons.length) {
      vectors.get(persons(i)).foreach { v =>
        idx ++= v.indices
        value ++= v.values
        rowPtr(i + 1) = v.indices.length.toLong
      }
      i += 1
    }
    i = 0
    while (i < persons.length) {
      rowPtr(i + 1) += rowPtr(i)
      i += 1
    }
    (rowPtr, idx.result(), value.result())
  }

  /** The reference's implementation, only under LOCREC_BACKEND=spark. */
  private lazy val sparkDelegate = new SparkKnnRecommender(
    placeRatingVectors, categoryRatingVectors, placeRatings, placeWeight, categoryWeight, kNearest)

  private lazy val (handle: Long, cleanable: java.lang.ref.Cleaner.Cleanable) =
    LocrecBackend.handleFor(this, LocrecBackend.KindKnn,
      LocrecBackend.framesKey(placeRatingVectors, categoryRatingVectors, placeRatings))(createIndex())

  /** A cache miss only: the native Parquet loader when the three frames are scans of local files, else collect -> CSR. */
  private def createIndex(): Long = {
    val paths = Seq(placeRatingVectors, categoryRatingVectors, placeRatings).map(LocrecBackend.localPathOf)
    if (paths.forall(_.isDefined))
      LocrecBackend.tryNativeLoad(LocrecNative.knnCreateFromParquet(paths(0).get, paths(1).get, paths(2).get)) match {
        case Some(h) => return h
        case None =>
      }
    createIndexByCollect()
  }

  /** collect -> CSR -> locrec_knn_create. */
  private def createIndexByCollect(): Long = {
    val (placeVectors, placeDim) = collectVectors(placeRatingVectors)
    val (categoryVectors, categoryDim) = collectVectors(categoryRatingVectors)
    val ratingRows = placeRatings
      .select(col("person_id").cast(LongType), col("place_id").cast(LongType), col("rating").cast(LongType))
      .collect()
    val ratingsByPerson: Map[Long, Array[Row]] = ratingRows.groupBy(_.getLong(0))
    val persons: Array[Long] = (placeVectors.keySet ++ categoryVectors.keySet ++ ratingsByPerson.keySet).toArray.sorted

    val (pRowPtr, pIdx, pVal) = toCsr(persons, placeVectors)
    val (cRowPtr, cIdx, cVal) = toCsr(persons, categoryVectors)
    val rRowPtr = new Array[Long](persons.length + 1)
    val rPlace = mutable.ArrayBuilder.make[Long]
    val rRating = mutable.ArrayBuilder.make[Long]
    var i = 0
    while (i < persons.length) {
      val rows = ratingsByPerson.getOrElse(persons(i), Array.empty[Row])
      rows.foreach { r =>
        rPlace += r.getLong(1)
        rRating += r.getLong(2)
      }
      rRowPtr(i + 1) = rRowPtr(i) + rows.length
      i += 1
    }
    LocrecNative.knnCreate(
      persons,
      pRowPtr, pIdx, pVal, placeDim,
      cRowPtr, cIdx, cVal, categoryDim,
      rRowPtr, rPlace.result(), rRating.result()
    )
  }

  def makeRecommendations(personId: Long): DataFrame = if (LocrecBackend.useSpark) sparkDelegate.makeRecommendations(personId) else LocrecBackend.lockOf(handle).synchronized {
    var places = new Array[Long](4096)
    var ratings = new Array[Double](4096)
    var count = LocrecNative.knnRecommend(handle, personId, placeWeight, categoryWeight, kNearest.toLong, places, ratings)
    if (count > places.length) { // the result has more rows than the first buffer: once more with room for all
      places = new Array[Long](count.toInt)
      ratings = new Array[Double](count.toInt)
      count = LocrecNative.knnRecommend(handle, personId, placeWeight, categoryWeight, kNearest.toLong, places, ratings)
    }
    val rows = (0 until count.toInt).map(i => Row(places(i), ratings(i)))
    spark.createDataFrame(spark.sparkContext.parallelize(rows, 1), outputSchema)
  }

  /** Additive (SURVEY.md 8b): makeRecommendations for many persons in one device pass -> (person_id, place_id, estimated_rating). */
  def makeRecommendationsBatch(personIds: Seq[Long]): DataFrame = LocrecBackend.lockOf(handle).synchronized {
    require(!LocrecBackend.useSpark, "makeRecommendationsBatch exists only in the gpu backend")
    val ids = personIds.toArray
    val offsets = new Array[Long](ids.length + 1)
    val needed = LocrecNative.knnRecommendBatch(handle, ids, placeWeight, categoryWeight, kNearest.toLong, offsets, null, null)
    val places = new Array[Long](math.max(needed, 1L).toInt)
    val ratings = new Array[Double](places.length)
    LocrecNative.knnRecommendBatch(handle, ids, placeWeight, categoryWeight, kNearest.toLong, offsets, places, ratings)
    val rows = ids.indices.flatMap { q =>
      (offsets(q).toInt until offsets(q + 1).toInt).map(i => Row(ids(q), places(i), ratings(i)))
    }
    val schema = StructType(StructField("person_id", LongType, nullable = false) +: outputSchema.fields)
    spark.createDataFrame(spark.sparkContext.parallelize(rows, 1), schema)
  }

  /** Drops this object's reference (idempotent); the device index stays cached for the next constructor. */
  override def close(): Unit = if (!LocrecBackend.useSpark) cleanable.clean()

}
