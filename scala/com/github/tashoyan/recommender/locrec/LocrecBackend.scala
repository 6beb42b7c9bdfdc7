package com.github.tashoyan.recommender.locrec

import java.lang.ref.Cleaner

import org.apache.hadoop.fs.Path
import org.apache.spark.sql.DataFrame
import org.apache.spark.sql.execution.datasources.LogicalRelation

/**
  * What the two operator classes share: the backend switch and the process-wide handle cache.
  *
  * Backend.  `LOCREC_BACKEND` (environment) or `-Dlocrec.backend=` selects `gpu` (default: liblocrec.so) or
  * `spark`: the reference's own DataFrame implementation, kept next to the replacement under the names
  * `SparkKnnRecommender` / `SparkStochasticRecommender` (INTEGRATION.md section 3: the maintainer renames the two
  * original classes, nothing else).  The switch is read once per JVM.
  *
  * Handle cache.  The reference's mains construct a new recommender from freshly read DataFrames for EVERY
  * request (KnnRecommenderMain.scala:53-67, StochasticRecommenderMain.scala:53-62) and close nothing.  The
  * device index therefore lives in the library's cache (include/locrec.h "Handle cache"), keyed by what the
  * DataFrames ARE - never by the weights, K, epsilon or maxIterations, which are arguments of every native
  * request call.  A recommender object holds a reference; `close()` or, for the unchanged mains that never
  * close, a `java.lang.ref.Cleaner` action drops it.  The handle itself stays on the device until the cache's
  * byte budget evicts it (least recently used first).
  */
object LocrecBackend {

  val KindKnn = 0
  val KindSg = 1

  lazy val useSpark: Boolean = {
    val v = sys.props.get("locrec.backend").orElse(sys.env.get("LOCREC_BACKEND")).getOrElse("gpu").trim.toLowerCase
    require(v == "gpu" || v == "spark", s"LOCREC_BACKEND must be 'gpu' or 'spark': $v")
    v == "spark"
  }

  private val cleaner: Cleaner = Cleaner.create()

  /**
    * Identity of a DataFrame that is a plain file scan (what `spark.read.parquet(fileName)` of the mains
    * returns): its input files, sorted, each with length and modification time - a builder that rewrites the
    * files invalidates the key.  Any other plan (filters, joins, in-memory rows as in the tests) has no cheap
    * identity: None, and the caller builds a private, uncached handle.
    */
  def frameKey(df: DataFrame): Option[String] = df.queryExecution.analyzed match {
    case _: LogicalRelation =>
      val files = df.inputFiles.sorted
      if (files.isEmpty) None
      else {
        val conf = df.sparkSession.sparkContext.hadoopConfiguration
        val parts = files.map { f =>
          val path = new Path(f)
          val status = path.getFileSystem(conf).getFileStatus(path)
          s"$f:${status.getLen}:${status.getModificationTime}"
        }
        Some(df.schema.catalogString + "#" + parts.mkString("|"))
      }
    case _ => None
  }

  /**
    * The local directory (or file) a plain file-scan DataFrame reads, when ALL its input files sit under one local
    * path - what `spark.read.parquet(fileName)` of the mains gives on a single host: the native Parquet loader
    * (LocrecNative.knnCreateFromParquet / sgCreateFromParquet) then builds the device handle straight from the files and
    * nothing is collected through the driver.  None for remote file systems, mixed parents or non-scan plans.
    */
  def localPathOf(df: DataFrame): Option[String] = df.queryExecution.analyzed match {
    case _: LogicalRelation =>
      val uris = df.inputFiles.map(new java.net.URI(_))
      if (uris.isEmpty || uris.exists(u => u.getScheme != null && u.getScheme != "file")) None
      else {
        val parents = uris.map(u => new java.io.File(u.getPath).getParent).distinct
        if (parents.length == 1) Some(parents.head) else None
      }
    case _ => None
  }

  /** Runs the native loader; None when the shim has no Parquet support (the caller then collects). */
  def tryNativeLoad(load: => Long): Option[Long] =
    try Some(load)
    catch { case _: UnsupportedOperationException | _: UnsatisfiedLinkError => None }

  /** All frames must have a key; one without makes the whole recommender uncacheable. */
  def framesKey(frames: DataFrame*): Option[String] = {
    val keys = frames.map(frameKey)
    if (keys.forall(_.isDefined)) Some(keys.flatten.mkString("||")) else None
  }

  /**
    * The device handle for `key`: from the cache, or `create` on a miss (published to the cache).  Without a key
    * the handle is private to `owner`.  Either way a Cleaner action releases it when `owner` becomes
    * unreachable; the returned Cleanable's `clean()` is what `close()` calls (idempotent).
    */
  def handleFor(owner: AnyRef, kind: Int, key: Option[String])(create: => Long): (Long, Cleaner.Cleanable) = {
    val handle = key match {
      case Some(k) =>
        val hit = LocrecNative.cacheAcquire(kind, k)
        if (hit != 0L) hit
        else {
          val before = LocrecNative.deviceBytesInUse()
          val created = create
          LocrecNative.cachePublish(kind, k, created, math.max(0L, LocrecNative.deviceBytesInUse() - before))
        }
      case None => create
    }
    // the action must not capture `owner` (it would never become unreachable): only the two primitives
    val cleanable = cleaner.register(owner, new ReleaseAction(kind, handle))
    (handle, cleanable)
  }

  private final class ReleaseAction(kind: Int, handle: Long) extends Runnable {
    // cacheRelease drops a reference of a cached handle and DESTROYS one that was never published
    override def run(): Unit = LocrecNative.cacheRelease(kind, handle)
  }

  /** One lock per device handle: a handle is one stream, users of a shared cached handle take turns. */
  private val locks = new java.util.concurrent.ConcurrentHashMap[java.lang.Long, Object]()

  def lockOf(handle: Long): Object =
    locks.computeIfAbsent(Long.box(handle), new java.util.function.Function[java.lang.Long, Object] {
      override def apply(h: java.lang.Long): Object = new Object
    })

}
