"""Import-path shim: ``from models.Encoder import Encoder`` (the reference's spelling) resolves to the MI355X mirror."""
