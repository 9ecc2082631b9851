"""Built artefacts (libcffm_hip.so, libcffm_libfm.so, the pybind11 layer); populated by `make`."""
