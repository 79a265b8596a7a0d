"""Drop-in module names of the reference (`utils.load_model`, `utils.model_utils`, `utils.reader`,
`utils.data_utils`, `utils.utils`) implemented on the MI355X engine (neuspeech1_amd)."""
