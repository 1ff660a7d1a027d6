"""Host-side helpers around the disparity stage: ``utils.geometry`` (Cassini / ERP re-projections, disparity -> depth, view
transform with the HIP z-buffer)."""
