"""TEST HARNESS, not product: a condensed stand-in for buffalo's own Python front (buffalo.algo ALS / BPRMF / WARP,
buffalo.data MatrixMarket / Stream, buffalo.misc.aux Option) that drives the C ABI through buffalo_amd.backend the way
stock buffalo would (same call order: /root/reference/buffalo/algo/{als,bpr,warp}.py).  It exists because the reference
package cannot be imported in this image (no h5py, no compiled extensions); in a real deployment stock buffalo's front
binds the library as INTEGRATION.md describes and none of this is needed.  It follows the reference's classes closely on
purpose -- it is what the drop-in tests replay -- and is deliberately kept out of the `buffalo_amd` package."""
