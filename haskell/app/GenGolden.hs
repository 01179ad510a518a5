{-# LANGUAGE BangPatterns #-}
-- | GenGolden — regenerate the golden vectors of tests/golden/ with the REAL reference.
--
-- WRITTEN BLIND: there is no GHC in the build image, so this program has not been compiled or
-- run.  It is the missing half of the oracle's pin (DESIGN.md section 2): the C++ oracle is pinned
-- by the reference's own four known-answer tests only, because nothing else can be had without
-- a Haskell toolchain.  Anyone who has one can close the gap:
--
--   1. python tests/golden/export_for_haskell.py          (writes tests/golden/*.in.txt)
--   2. in a checkout of ocramz/rp-tree v0.7.1: copy this file to app/GenGolden.hs and build it
--      with the library's sources in scope (Data.RPTree.Internal is not an exposed module):
--        ghc -O1 -isrc app/GenGolden.hs -o gen-golden
--   3. ./gen-golden < tests/golden/forest_dense_1000x16.in.txt > forest_dense_1000x16.hs.txt
--   4. python tests/golden/diff_haskell.py forest_dense_1000x16.hs.txt
--      compares perm / thr / margins / candidates / knn ids+distances / recallWith, bit for bit,
--      with the committed fixture (= what the oracle and the device produce).
--
-- No random generator is involved: data, hyperplanes (dense-ified, zeros = absent components)
-- and queries are all read from the input, so splitmix's draw order does not matter.
--
-- Input (whitespace separated): n d T L minLeaf nq k, then X[n][d], R[T][L][d], Q[nq][d].
-- Output: one record per line, see the `put*` functions.
module Main (main) where

import qualified Data.IntMap.Strict as IM
import qualified Data.Vector as V
import qualified Data.Vector.Unboxed as VU
import Data.Foldable (toList)

import Data.RPTree (candidates, knn, recallWith, Embed(..), fromVectorDv, fromVectorSv, metricL2)
import Data.RPTree.Internal (RPT(..), RPTree(..), createMulti, getMargin, SVector, DVector)

type Pt = Embed DVector Double Int      -- payload = the point's index in the input

main :: IO ()
main = do
  ws <- words <$> getContents
  let (hdr, rest0) = splitAt 7 ws
      [n, d, nt, l, minLeaf, nq, k] = map read hdr :: [Int]
      (xs, rest1) = splitAt (n * d) rest0
      (rs, rest2) = splitAt (nt * l * d) rest1
      qs = take (nq * d) rest2
      rowsOf m ys = [ VU.fromList (map read (take d (drop (i * d) ys))) | i <- [0 .. m - 1] ] :: [VU.Vector Double]
      dat = V.fromList [ Embed (fromVectorDv v) i | (i, v) <- zip [0 ..] (rowsOf n xs) ] :: V.Vector Pt
      -- hyperplanes back to SVector: nonzero components, ascending index (Internal.hs:99-105)
      sparseOf v = fromVectorSv d (VU.filter ((/= 0) . snd) (VU.indexed v)) :: SVector Double
      rvss = IM.fromList [ (t, V.fromList [ sparseOf r | r <- take l (drop (t * l) (rowsOf (nt * l) rs)) ])
                         | t <- [0 .. nt - 1] ]
      queries = map fromVectorDv (rowsOf nq qs)
      rpts = createMulti l minLeaf rvss dat                       -- Internal.hs:227-240
      forest = IM.intersectionWith RPTree rvss rpts                -- as Batch.hs:63
  mapM_ (\(t, tr) -> putTree l t tr) (IM.toList rpts)
  mapM_ (\(qi, q) -> do
           mapM_ (\(t, tr) -> putLine ("cand " ++ show qi ++ " " ++ show t)
                                       (map (show . eData) (toList (candidates tr q))))
                 (IM.toList forest)
           let res = knn metricL2 k forest q                       -- RPTree.hs:168-176
           putLine ("knn_ids " ++ show qi) (map (show . eData . snd) (toList res))
           putLine ("knn_dist " ++ show qi) (map (show . fst) (toList res))
           putLine ("recall " ++ show qi) [show (recallWith metricL2 forest k q :: Double)])
        (zip [0 :: Int ..] queries)

putLine :: String -> [String] -> IO ()
putLine tag xs = putStrLn (unwords (tag : xs))

-- | perm (leaves left to right) and the node arrays in heap order (root 0, children 2h+1, 2h+2);
-- slots that are not a Bin print as "nan" — the flat layout of include/rptree_hip.h.
putTree :: Int -> Int -> RPT Double () (V.Vector Pt) -> IO ()
putTree l t tr = do
  putLine ("perm " ++ show t) (map (show . eData) (concatMap toList (leavesLR tr)))
  let nodes = 2 ^ l - 1 :: Int
      bins = IM.fromList (go 0 tr)
      col f = [ maybe "nan" (show . f) (IM.lookup h bins) | h <- [0 .. nodes - 1] ]
  putLine ("thr " ++ show t) (col (\(a, _, _) -> a))
  putLine ("mglo " ++ show t) (col (\(_, b, _) -> b))
  putLine ("mghi " ++ show t) (col (\(_, _, c) -> c))
  where
    go :: Int -> RPT Double () a -> [(Int, (Double, Double, Double))]
    go h (Bin _ thr mg ll rr) = let (lo, hi) = getMargin mg
                                in (h, (thr, lo, hi)) : go (2 * h + 1) ll ++ go (2 * h + 2) rr
    go _ (Tip _ _) = []
    leavesLR (Bin _ _ _ ll rr) = leavesLR ll ++ leavesLR rr
    leavesLR (Tip _ xs) = [xs]
