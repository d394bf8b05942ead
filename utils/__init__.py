"""Import-path shim: the reference's ``utils.load_dataset`` / ``utils.eval_utils`` / ``utils.utils`` spellings resolve to
the MI355X build (``lstc_vad_amd``)."""
