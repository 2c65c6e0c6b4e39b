"""MI355X-native (gfx950) implementation of CMR-Agent's feature-extraction, matching and
agent-iteration hot path.  See DESIGN.md.  Importing the package does not load the HIP
library; the first op call does (cmr_agent_amd._lib) and fails loudly if it is missing."""
__version__ = "0.1.0"
