"""``ops`` -- host-side mirror of the reference's /root/reference/detection/ops package
(functions.MSDeformAttnFunction, modules.MSDeformAttn) on top of the gfx950 HIP library.
Importable as a top-level package when ``vit-adapter_amd/`` is on sys.path, exactly like the
reference's ``ops`` directory is when ``detection/`` or ``segmentation/`` is the cwd."""
