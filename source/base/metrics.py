from ppsurf_amd.lightning_api import (calc_accuracy, calc_precision, calc_recall, calc_f1,  # noqa: F401
                                      compare_predictions_binary_tensors)
