{-# LANGUAGE ForeignFunctionInterface #-}
{-# LANGUAGE FlexibleContexts #-}
-- | Binding of the MI355X hot path (include/rptree_hip.h) under the unchanged Data.RPTree API.
--
-- WRITTEN BLIND: there is no GHC in the build image, so this module has not been compiled.
-- It shows the binding a maintainer would add next to src/Data/RPTree/Batch.hs; the tested
-- mirrors of exactly this call sequence are rp-tree_amd/python/rptree_amd/__init__.py and
-- rp-tree_amd/host/rptree.hpp.
module Data.RPTree.HIP (forestBatchHIP, forestBatchHIPWith, forestHIP, withDeviceData, withDeviceForest,
                        withDeviceForestOn, knnHIP, ProjMode(..), FlatForest(..), DeviceForest(..), DeviceData(..)) where

import Control.Exception (Exception, bracket, throwIO)
import Control.Monad (when)
import Data.Int (Int8, Int32, Int64)
import Data.Word (Word64)
import Foreign.C.String (CString, peekCString)
import Foreign.Marshal.Alloc (alloca)
import Foreign.Ptr (Ptr)
import Foreign.Storable (peek)
import System.IO.Unsafe (unsafePerformIO)
import qualified Data.IntMap.Strict as IM
import qualified Data.Vector as V
import qualified Data.Vector.Storable as VS
import qualified Data.Vector.Storable.Mutable as VSM
import qualified Data.Vector.Unboxed as VU

import System.Random.SplitMix.Distributions (sample, stdNormal)
import Data.RPTree.Gen (sparse)
import Data.RPTree.Internal (Embed(..), DVector(..), SVector(..), RPForest, RPTree(..), RPT(..), Margin(..))
import Data.Semigroup (Max(..), Min(..))

data Ctx; data Dataset; data Forest; data Comm; data ShardedForest

-- `safe`: every call may launch kernels and synchronise; do not block the RTS.
foreign import ccall safe "rpt_ctx_create"          c_ctx_create     :: Int32 -> Ptr (Ptr Ctx) -> IO Int32
foreign import ccall safe "rpt_ctx_destroy"         c_ctx_destroy    :: Ptr Ctx -> IO Int32
foreign import ccall safe "rpt_dataset_dense_host"  c_dataset_dense  :: Ptr Ctx -> Ptr Double -> Int64 -> Int32 -> Int32 -> Ptr (Ptr Dataset) -> IO Int32
foreign import ccall safe "rpt_dataset_free"        c_dataset_free   :: Ptr Dataset -> IO Int32
foreign import ccall safe "rpt_forest_build"        c_forest_build   :: Ptr Ctx -> Ptr Dataset -> Ptr Double -> Int32 -> Int32 -> Int32 -> Int32 -> Ptr (Ptr Forest) -> IO Int32
foreign import ccall safe "rpt_forest_free"         c_forest_free    :: Ptr Forest -> IO Int32
foreign import ccall safe "rpt_forest_get_perm"     c_forest_perm    :: Ptr Forest -> Ptr Int32 -> IO Int32
foreign import ccall safe "rpt_forest_get_nodes"    c_forest_nodes   :: Ptr Forest -> Ptr Double -> Ptr Double -> Ptr Double -> IO Int32
-- forest / tree (Conduit.hs:58-121): the fold of insert over chunks, Internal.hs:245-297
foreign import ccall safe "rpt_forest_stream_build" c_stream_build   :: Ptr Ctx -> Ptr Dataset -> Ptr Double -> Int32 -> Int32 -> Int32 -> Int64 -> Int32 -> Ptr (Ptr Forest) -> IO Int32
foreign import ccall safe "rpt_forest_get_topology" c_forest_topo    :: Ptr Forest -> Ptr Int64 -> Ptr Int8 -> Ptr Int64 -> Ptr Int64 -> Ptr Int64 -> Ptr Int64 -> IO Int32
foreign import ccall safe "rpt_knn_host"            c_knn_host       :: Ptr Ctx -> Ptr Forest -> Ptr Dataset -> Ptr Dataset -> Int32 -> Int32 -> Ptr Int32 -> Ptr Double -> Ptr Int32 -> IO Int32
-- knnH (RPTree.hs:199-217): two calls, the first with null outputs returns the result size
foreign import ccall safe "rpt_knnh_host"           c_knnh_host      :: Ptr Ctx -> Ptr Forest -> Ptr Dataset -> Ptr Dataset -> Int32 -> Ptr Int64 -> Ptr Int32 -> Ptr Double -> Int64 -> Ptr Int64 -> IO Int32
foreign import ccall unsafe "rpt_last_error"        c_last_error     :: IO CString
-- multi-GPU (csrc/comm.hip on librccl): one process drives n devices; per-device arguments are
-- arrays with one entry per device (Foreign.Marshal.Array.withArray)
foreign import ccall safe "rpt_comm_init"            c_comm_init      :: Int32 -> Ptr (Ptr Comm) -> IO Int32
foreign import ccall safe "rpt_comm_destroy"         c_comm_destroy   :: Ptr Comm -> IO Int32
foreign import ccall safe "rpt_comm_ctx"             c_comm_ctx       :: Ptr Comm -> Int32 -> Ptr (Ptr Ctx) -> IO Int32
foreign import ccall safe "rpt_forest_build_sharded" c_build_sharded  :: Ptr Comm -> Ptr (Ptr Dataset) -> Ptr Double -> Int32 -> Int32 -> Int32 -> Int32 -> Ptr (Ptr ShardedForest) -> IO Int32
foreign import ccall safe "rpt_sharded_forest_free"  c_sharded_free   :: Ptr ShardedForest -> IO Int32
foreign import ccall safe "rpt_knn_sharded"          c_knn_sharded    :: Ptr Comm -> Ptr ShardedForest -> Ptr (Ptr Dataset) -> Ptr (Ptr Dataset) -> Int32 -> Int32 -> Ptr Int32 -> Ptr Double -> Ptr Int32 -> IO Int32

-- | Projection kernel of a build (include/rptree_hip.h RPT_PROJ_*).  'ProjAuto' on Double data is the
-- exact-order kernel (separate multiply and add in innerSD's order, Internal.hs:369-382: leaf
-- assignments identical to the pure library's); 'ProjMfma' is the matrix-core kernel bench.py times
-- (projection values within 1e-5 |x||r|, a handful of leaf flips per million points at most).
data ProjMode = ProjAuto | ProjExact | ProjMfma deriving (Eq, Show)

projFlag :: ProjMode -> Int32
projFlag ProjAuto = 0
projFlag ProjExact = 1
projFlag ProjMfma = 2

newtype RPTHipError = RPTHipError String deriving Show
instance Exception RPTHipError          -- next to RPTError (Internal.hs:66-72)

check :: Int32 -> IO ()
check 0 = pure ()
check _ = c_last_error >>= peekCString >>= throwIO . RPTHipError

-- | The flat forest of include/rptree_hip.h, copied out so ordinary 'RPTree' values can be rebuilt.
data FlatForest = FlatForest
  { ffPerm :: VS.Vector Int32, ffThr, ffLo, ffHi :: VS.Vector Double
  , ffN, ffTrees, ffDepth, ffMinLeaf :: Int }

-- | Dense-ify one hyperplane in O(d + nnz): zeros where the sparse vector has no component.
denseOf :: Int -> SVector Double -> VS.Vector Double
denseOf dim (SV _ vv) = VS.replicate dim 0 VS.// VU.toList vv

-- | Drop-in for 'Data.RPTree.Batch.forestBatch' (Batch.hs:48-63) on dense data.  Everything the
-- result needs is copied out, so the device objects are released before returning (bracket:
-- also when a call fails) — nothing is left to finalisers.
forestBatchHIP :: Word64 -> Int -> Int -> Int -> Double -> Int
               -> V.Vector (Embed DVector Double x)
               -> RPForest Double (V.Vector (Embed DVector Double x))
forestBatchHIP = forestBatchHIPWith ProjAuto

-- | ... with the projection kernel chosen by the caller ('ProjMfma' = the timed one).  One context and
-- one upload per call: a host that builds several forests over the same points, or queries the
-- forest afterwards, uses 'withDeviceData' / 'withDeviceForestOn' instead and pays the upload once.
forestBatchHIPWith :: ProjMode -> Word64 -> Int -> Int -> Int -> Double -> Int
                   -> V.Vector (Embed DVector Double x)
                   -> RPForest Double (V.Vector (Embed DVector Double x))
forestBatchHIPWith mode seed maxd minl ntrees pnz dim src = unsafePerformIO $ do
  -- hyperplanes: sampled by the HOST exactly as Batch.hs:59-61 does
  let rvss = sample seed $ V.replicateM ntrees (V.replicateM maxd (sparse pnz dim stdNormal))
      rflat = VS.concat [ denseOf dim r | rvs <- V.toList rvss, r <- V.toList rvs ]     -- R[T][L][d]
      xflat = VS.concat [ VS.convert v | Embed (DV v) _ <- V.toList src ]                -- X[N][d]
      n = V.length src
      nodes = 2 ^ maxd - 1
      acquire mk = alloca $ \pp -> mk pp >>= check >> peek pp
  ff <- bracket (acquire (c_ctx_create 0)) c_ctx_destroy $ \ctx ->
        bracket (acquire (\pp -> VS.unsafeWith xflat (\px -> c_dataset_dense ctx px (fromIntegral n) (fromIntegral dim) 0 pp)))
                c_dataset_free $ \ds ->
        bracket (acquire (\pp -> VS.unsafeWith rflat (\pr -> c_forest_build ctx ds pr (fromIntegral ntrees) (fromIntegral maxd) (fromIntegral minl) (projFlag mode) pp)))
                c_forest_free $ \f -> do
          perm <- VSM.new (ntrees * n); thr <- VSM.new (ntrees * nodes); lo <- VSM.new (ntrees * nodes); hi <- VSM.new (ntrees * nodes)
          VSM.unsafeWith perm (c_forest_perm f) >>= check
          VSM.unsafeWith thr (\a -> VSM.unsafeWith lo (\b -> VSM.unsafeWith hi (c_forest_nodes f a b))) >>= check
          FlatForest <$> VS.freeze perm <*> VS.freeze thr <*> VS.freeze lo <*> VS.freeze hi
                     <*> pure n <*> pure ntrees <*> pure maxd <*> pure minl
  pure $ IM.fromList [ (t, RPTree (rvss V.! t) (rebuild ff src t)) | t <- [0 .. ntrees - 1] ]

-- | Rebuild the lazy 'RPT' (Internal.hs:139-149) of tree t from the flat arrays: topology is a
-- pure function of (N, minLeaf, maxDepth) (Internal.hs:289,495,503).
rebuild :: FlatForest -> V.Vector a -> Int -> RPT Double () (V.Vector a)
rebuild ff src t = go 0 0 0 (ffN ff)
  where
    nodes = 2 ^ ffDepth ff - 1
    go lev h off m
      | lev >= ffDepth ff || m <= ffMinLeaf ff =
          Tip () (V.generate m (\i -> src V.! fromIntegral (ffPerm ff VS.! (t * ffN ff + off + i))))
      | otherwise =
          let nh = m `div` 2; ix = t * nodes + h
          in Bin () (ffThr ff VS.! ix) (Margin (Max (ffLo ff VS.! ix)) (Min (ffHi ff VS.! ix)))
                 (go (lev + 1) (2 * h + 1) off nh) (go (lev + 1) (2 * h + 2) (off + nh) (m - nh))

-- | Drop-in for 'Data.RPTree.Conduit.forest' (Conduit.hs:104-121) on dense data once the conduit
-- has been sunk into a vector (@src <- runConduit (source .| C.sinkVector)@): the reference's fold
-- of 'insert' over chunks of @chunk@ points runs on the device (rpt_forest_stream_build) and the
-- explicit topology it returns (kind 1 = Bin, 2 = Tip per heap slot, the same for every tree) is
-- turned back into ordinary 'RPT' values.
forestHIP :: Word64 -> Int -> Int -> Int -> Int -> Double -> Int
          -> V.Vector (Embed DVector Double x)
          -> RPForest Double (V.Vector (Embed DVector Double x))
forestHIP seed maxd minl ntrees chunk pnz dim src = unsafePerformIO $ do
  let rvss = sample seed $ V.replicateM ntrees (V.replicateM maxd (sparse pnz dim stdNormal))  -- Conduit.hs:116-118
      rflat = VS.concat [ denseOf dim r | rvs <- V.toList rvss, r <- V.toList rvs ]
      xflat = VS.concat [ VS.convert v | Embed (DV v) _ <- V.toList src ]
      n = V.length src
      slots = 2 ^ (maxd + 1) - 1
      acquire mk = alloca $ \pp -> mk pp >>= check >> peek pp
  (perm, thr, lo, hi, kind, loff, llen) <-
    bracket (acquire (c_ctx_create 0)) c_ctx_destroy $ \ctx ->
    bracket (acquire (\pp -> VS.unsafeWith xflat (\px -> c_dataset_dense ctx px (fromIntegral n) (fromIntegral dim) 0 pp)))
            c_dataset_free $ \ds ->
    bracket (acquire (\pp -> VS.unsafeWith rflat (\pr -> c_stream_build ctx ds pr (fromIntegral ntrees) (fromIntegral maxd) (fromIntegral minl) (fromIntegral chunk) 0 pp)))
            c_forest_free $ \f -> do
      perm <- VSM.new (ntrees * n); thr <- VSM.new (ntrees * slots); lo <- VSM.new (ntrees * slots); hi <- VSM.new (ntrees * slots)
      kind <- VSM.new slots; loff <- VSM.new slots; llen <- VSM.new slots
      VSM.unsafeWith perm (c_forest_perm f) >>= check
      VSM.unsafeWith thr (\a -> VSM.unsafeWith lo (\b -> VSM.unsafeWith hi (c_forest_nodes f a b))) >>= check
      alloca $ \ps -> alloca $ \ph -> alloca $ \pd ->
        VSM.unsafeWith kind (\k -> VSM.unsafeWith loff (\o -> VSM.unsafeWith llen (\l -> c_forest_topo f ps k o l ph pd))) >>= check
      (,,,,,,) <$> VS.freeze perm <*> VS.freeze thr <*> VS.freeze lo <*> VS.freeze hi
               <*> VS.freeze kind <*> VS.freeze loff <*> VS.freeze llen
  let tree t = go 0
        where
          go h | kind VS.! h == 1 =
                   let ix = t * slots + h
                   in Bin () (thr VS.! ix) (Margin (Max (lo VS.! ix)) (Min (hi VS.! ix))) (go (2 * h + 1)) (go (2 * h + 2))
               | otherwise =            -- Tip (kind 2; an absent slot never hangs below a Bin)
                   Tip () (V.generate (fromIntegral (llen VS.! h))
                             (\i -> src V.! fromIntegral (perm VS.! (t * n + fromIntegral (loff VS.! h) + i))))
  pure $ IM.fromList [ (t, RPTree (rvss V.! t) (tree t)) | t <- [0 .. ntrees - 1] ]

-- | A forest that STAYS on the device, for hosts that build once and query many times (what the
-- north star means by "host code stays Haskell": the handle is the currency, as 'RPForest' is in
-- the reference).  The context, the packed dataset and the forest live for the extent of the
-- callback; queries go through 'knnHIP' without re-uploading anything.
data DeviceForest = DeviceForest { dfCtx :: Ptr Ctx, dfData :: Ptr Dataset, dfForest :: Ptr Forest }

-- | A context and the packed point set on the device, for the extent of the callback: the upload
-- (1 GB at C2: 18 ms over PCIe against a 5 ms build) is paid once however many forests are built.
data DeviceData = DeviceData { ddCtx :: Ptr Ctx, ddData :: Ptr Dataset, ddDim :: Int }

withDeviceData :: Int -> V.Vector (Embed DVector Double x) -> (DeviceData -> IO a) -> IO a
withDeviceData dim src act = do
  let xflat = VS.concat [ VS.convert v | Embed (DV v) _ <- V.toList src ]
      acquire mk = alloca $ \pp -> mk pp >>= check >> peek pp
  bracket (acquire (c_ctx_create 0)) c_ctx_destroy $ \ctx ->
    bracket (acquire (\pp -> VS.unsafeWith xflat (\px -> c_dataset_dense ctx px (fromIntegral (V.length src)) (fromIntegral dim) 0 pp)))
            c_dataset_free $ \ds -> act (DeviceData ctx ds dim)

-- | One forest over device-resident points (any number of these per 'withDeviceData').
withDeviceForestOn :: DeviceData -> ProjMode -> Word64 -> Int -> Int -> Int -> Double
                   -> (DeviceForest -> IO a) -> IO a
withDeviceForestOn (DeviceData ctx ds dim) mode seed maxd minl ntrees pnz act = do
  let rvss = sample seed $ V.replicateM ntrees (V.replicateM maxd (sparse pnz dim stdNormal))
      rflat = VS.concat [ denseOf dim r | rvs <- V.toList rvss, r <- V.toList rvs ]
      acquire mk = alloca $ \pp -> mk pp >>= check >> peek pp
  bracket (acquire (\pp -> VS.unsafeWith rflat (\pr -> c_forest_build ctx ds pr (fromIntegral ntrees) (fromIntegral maxd) (fromIntegral minl) (projFlag mode) pp)))
          c_forest_free $ \f -> act (DeviceForest ctx ds f)

withDeviceForest :: Word64 -> Int -> Int -> Int -> Double -> Int -> V.Vector (Embed DVector Double x)
                 -> (DeviceForest -> IO a) -> IO a
withDeviceForest seed maxd minl ntrees pnz dim src act =
  withDeviceData dim src $ \dd -> withDeviceForestOn dd ProjAuto seed maxd minl ntrees pnz act

-- | 'knn metricL2 k' (RPTree.hs:168-176) for a batch of dense queries: ids and distances.
knnHIP :: Ptr Ctx -> Ptr Forest -> Ptr Dataset -> Ptr Dataset -> Int -> Int -> IO (VS.Vector Int32, VS.Vector Double, VS.Vector Int32)
knnHIP ctx f ds qs nq k = do
  ids <- VSM.new (nq * k); dist <- VSM.new (nq * k); cnt <- VSM.new nq
  VSM.unsafeWith ids (\a -> VSM.unsafeWith dist (\b -> VSM.unsafeWith cnt (c_knn_host ctx f ds qs (fromIntegral k) 0 a b))) >>= check
  (,,) <$> VS.freeze ids <*> VS.freeze dist <*> VS.freeze cnt
