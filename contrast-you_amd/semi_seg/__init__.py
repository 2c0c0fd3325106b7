"""MI355X-native restatement of the `semi_seg` package surface used by the
SemiSupervisedEpocher + InfoNCE hot path (same module and symbol names as the reference)."""
