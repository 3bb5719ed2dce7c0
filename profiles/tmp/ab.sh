for d in 0 1 2 3 4 5 6 7; do AXT_DBG=$d python profiles/cnn_kernels.py; done
