"""MI355X-native restatement of the `contrastyou` package surface used by the
SemiSupervisedEpocher + InfoNCE hot path (same module and symbol names as the reference)."""
